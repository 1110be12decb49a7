#!/bin/bash
# k_wide after the cheap flush redo paths: parity (all wide tests incl. many partitions), digests, C3 full size
out=gpurun_out/${1:-w11}; mkdir -p $out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py tests/test_reservations.py -x -q -m gpu -k "wide" > $out/parity.log 2>&1
echo "parity rc=$?" >> $out/parity.log
timeout 400 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide" > $out/fullrun.log 2>&1
echo "fullrun rc=$?" >> $out/fullrun.log
timeout 300 python tools/c3_compare.py wide 2>&1 | grep -v amdgpu.ids > $out/c3.txt
CNS_SELECT_KERNEL=wide CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so timeout 200 python tools/prof_wide.py C3 2>&1 | grep -v amdgpu.ids >> $out/c3.txt
grep -v amdgpu.ids $out/parity.log | tail -n 3; grep -v amdgpu.ids $out/fullrun.log | tail -n 14; cat $out/c3.txt
