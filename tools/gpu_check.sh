#!/bin/bash
# usage: tools/gpu_check.sh <outdir> [pytest -k expr for the full-run cases]   — pipe-kernel parity + full-run digests + cycle breakdown
out=gpurun_out/$1; mkdir -p $out
sel=${2:-"c2 or c4 or c5 or tile1 or tile3"}
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py tests/test_reservations.py -x -q -m gpu -k pipe > $out/parity.log 2>&1
echo "parity rc=$?" >> $out/parity.log
timeout 400 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "pipe and ($sel)" > $out/fullrun_pipe.log 2>&1
echo "fullrun rc=$?" >> $out/fullrun_pipe.log
for cfg in C5 C4; do
CNS_SELECT_KERNEL=pipe CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so timeout 120 python tools/prof_pipe.py $cfg 2>&1 | grep -v amdgpu.ids > $out/prof_$cfg.txt
done
tail -n 4 $out/parity.log; grep -v amdgpu.ids $out/fullrun_pipe.log | tail -n 14; cat $out/prof_*.txt
