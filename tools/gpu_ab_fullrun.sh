#!/bin/bash
# A/B on ONE box: tools/gpu_ab_fullrun.sh <out> <pytest -k expression> <lib|default>... — the full-size digests (parity) and their
# timings under each library, k_wide without the retry, two passes
export TMPDIR=/tmp
out=gpurun_out/$1; sel=$2; shift; shift
mkdir -p $(dirname $out); : > $out
for rep in 1 2; do
for lib in "$@"; do
  [ "$lib" = default ] && unset CNS_ENGINE_LIB || export CNS_ENGINE_LIB=$lib
  echo "== lib: $lib (pass $rep)" >> $out
  timeout 600 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "$sel" 2>&1 | grep "identical\|passed\|failed\|rror\|differs" | sed 's/identical to the oracle.*; k_wide/k_wide/' | cut -c1-200 >> $out
done
done
cat $out
