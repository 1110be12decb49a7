#!/bin/bash
# k_wide with the same-XCD exchange: parity, full-run digests, then the FAR variant and the cycle breakdown
out=gpurun_out/w9; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py tests/test_reservations.py -x -q -m gpu -k wide > $out/parity.log 2>&1
echo "parity rc=$?" >> $out/parity.log
timeout 300 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide and (c2 or c4 or c5 or tile19 or tile37)" > $out/fullrun.log 2>&1
echo "fullrun rc=$?" >> $out/fullrun.log
CNS_ENGINE_LIB=build_var/v_far.so timeout 300 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide and (c4 or c5 or tile19)" > $out/fullrun_far.log 2>&1
echo "fullrun far rc=$?" >> $out/fullrun_far.log
for cfg in "C4" "C5" "C3 130000 8192 1"; do
CNS_SELECT_KERNEL=wide CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so timeout 200 python tools/prof_wide.py $cfg 2>&1 | grep -v amdgpu.ids >> $out/prof.txt
done
grep -v amdgpu.ids $out/parity.log | tail -n 8; grep -v amdgpu.ids $out/fullrun.log | tail -n 12; grep -v amdgpu.ids $out/fullrun_far.log | tail -n 8; cat $out/prof.txt
