#!/bin/bash
# A/B on ONE box: the default library against build_var/v_<tag>.so for every tag given; full-size runs (parity vs the digests)
export TMPDIR=/tmp
for rep in 1 2; do
for v in "" "$@"; do
  lib=""; [ -n "$v" ] && lib=build_var/v_$v.so
  echo "== lib: ${v:-default} (pass $rep)"
  CNS_ENGINE_LIB=$lib timeout 400 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide and not wide32 and (c4 or c5 or c3 or c2 or tile19)" 2>&1 | grep "identical\|passed\|failed\|rror" | sed 's/identical to the oracle.*; k_wide/k_wide/'
done
done
