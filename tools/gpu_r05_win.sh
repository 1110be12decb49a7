#!/bin/bash
# round 5, windows of jobs per exchange: small-case parity under k_wide (64 waves), then the full-size digests with the window off / on
export TMPDIR=/tmp
out=gpurun_out/${1:-r05a}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_reservations.py -q -m gpu -x -k "wide and not wide32" > $out/parity.log 2>&1; tail -5 $out/parity.log
for w in 4 0 2 3; do
  echo "== CNS_WIDE_WINDOW=$w"
  CNS_WIDE_WINDOW=$w timeout 600 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide and not wide32 and (c2 or c4] or c5 or c4r])" 2>&1 | grep "identical\|passed\|failed\|rror\|differs" | sed 's/identical to the oracle.*; k_wide/k_wide/' | tee -a $out/fullrun_w$w.log
done
