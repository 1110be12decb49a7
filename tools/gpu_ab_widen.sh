export TMPDIR=/tmp
for cfg in C4 C5 C2; do
for lib in build_var/v_base.so cranesched_amd/libcrane_gpu_nodeselect.so; do
  echo "== $cfg wide $lib"
  CNS_SELECT_KERNEL=wide CNS_ENGINE_LIB=$lib timeout 120 python tools/prof_wide.py $cfg 2>&1 | head -2
done; done
for k in pipe legacy; do
for lib in build_var/v_base.so cranesched_amd/libcrane_gpu_nodeselect.so; do
  echo "== C4 $k $lib"
  CNS_SELECT_KERNEL=$k CNS_ENGINE_LIB=$lib timeout 120 python tools/prof_wide.py C4 2>&1 | head -1
done; done
echo "== parity"
timeout 500 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -x -k "c4 or c2 or tile" 2>&1 | tail -3
