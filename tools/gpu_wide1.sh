#!/bin/bash
# first runs of k_wide: parity tests, full-run digests (with timings), bench line
out=gpurun_out/w1; mkdir -p $out
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py tests/test_reservations.py -x -q -m gpu -k wide > $out/parity.log 2>&1
echo "parity rc=$?" >> $out/parity.log
timeout 500 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide" > $out/fullrun.log 2>&1
echo "fullrun rc=$?" >> $out/fullrun.log
CNS_SELECT_KERNEL=wide timeout 300 python bench.py --steps 3 --warmup 1 > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
grep -v amdgpu.ids $out/parity.log | tail -n 30; grep -v amdgpu.ids $out/fullrun.log | tail -n 30; cat $out/bench.json; tail -5 $out/bench.err
