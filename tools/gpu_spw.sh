#!/bin/bash
# A/B of the scanner-wave count / placement of k_wide: default library (8 workgroups x 4 waves) vs variants
export TMPDIR=/tmp
for v in build_var/v_w16.so build_var/v_w16b.so build_var/v_w64.so; do
echo "== parity (small cases) with $v"
CNS_ENGINE_LIB=$v CNS_SELECT_KERNEL=wide timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "wide" 2>&1 | tail -2
done
for v in "" build_var/v_w16.so build_var/v_w16b.so build_var/v_w64.so; do
  echo "== lib: ${v:-default}"
  CNS_ENGINE_LIB=$v timeout 400 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide and (c4 or c5 or c3 or c2 or tile19)" 2>&1 | grep "identical\|passed\|failed\|rror" | sed 's/identical to the oracle.*; k_wide/k_wide/'
done
