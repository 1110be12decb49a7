#!/bin/bash
export TMPDIR=/tmp
for v in "" build_var/v_nap3.so build_var/v_nap0.so; do
  echo "== lib: ${v:-default}"
  CNS_ENGINE_LIB=$v timeout 300 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide and (c4 or c5)" 2>&1 | grep "identical" | sed 's/identical to the oracle.*; k_wide/k_wide/'
done
