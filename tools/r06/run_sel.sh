# k_select builds against the whole-queue digests: tools/r06/run_sel.sh <outdir> lib ...   (lib: build_var/v_<lib>.so; "-" = the installed library)
out=gpurun_out/$1; mkdir -p $out; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset CNS_ENGINE_LIB; else export CNS_ENGINE_LIB=build_var/v_$v.so; fi
  echo "== $v legacy" | tee -a $out/sel.txt
  CNS_SELECT_KERNEL=legacy timeout 1500 python -m pytest tests/test_gpu_fullrun.py -x -q -k "${LEGACY_K:-not c3 and not 64k}" -s 2>&1 | grep -E "identical|differs|passed|failed|Error|fault" | cut -c1-230 | tee -a $out/sel.txt
  echo "== $v wide" | tee -a $out/sel.txt
  CNS_SELECT_KERNEL=wide timeout 900 python -m pytest tests/test_gpu_fullrun.py -x -q -k "c4rp or c4all" -s 2>&1 | grep -E "identical|differs|passed|failed|Error|fault" | cut -c1-230 | tee -a $out/sel.txt
done
