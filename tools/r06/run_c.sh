out=gpurun_out/${1:-r06x}; mkdir -p $out
export CNS_WIDE_NO_RETRY=1
for c in ${CFGS:-C5 C4 C2}; do
  timeout 300 python tools/var_bench.py $c 2>&1 | grep -v amdgpu.ids | tee -a $out/bench.txt
done
CNS_SELECT_KERNEL=wide timeout 1500 python -m pytest tests/test_gpu_fullrun.py -x -q -k "wide and not c4rp and not c4all and not c3" -s 2>&1 | grep -E "identical|differs|passed|failed|Error|fault" | cut -c1-260 | tee $out/fullrun_wide.log
