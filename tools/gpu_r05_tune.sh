export TMPDIR=/tmp
out=gpurun_out/r05k; mkdir -p $out
for v in "" j24 m3 m4; do
  lib=""; [ -n "$v" ] && lib=build_var/v_$v.so
  w=16; [ -n "$v" ] && w=24
  echo "== lib ${v:-default} CNS_WIDE_WINDOW=$w"
  CNS_ENGINE_LIB=$lib CNS_WIDE_WINDOW=$w timeout 600 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide and not wide32 and (c2 or c4] or c5 or c4r])" 2>&1 | grep "identical\|passed\|failed\|rror\|differs" | sed 's/identical to the oracle.*; k_wide/k_wide/' | tee -a $out/fullrun_${v:-default}.log
done
