export TMPDIR=/tmp
out=gpurun_out/r05i; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_reservations.py tests/test_gpu_windows.py tests/test_overlap.py tests/test_gpu_vs_reference.py -q -m gpu -x > $out/parity.log 2>&1; tail -3 $out/parity.log
for t in 1 0; do for w in 16 0; do
  echo "== CNS_WIDE_TESTER_OPT=$t CNS_WIDE_WINDOW=$w"
  CNS_WIDE_TESTER_OPT=$t CNS_WIDE_WINDOW=$w timeout 600 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide and not wide32 and (c2 or c3 or c4] or c5 or c4r])" 2>&1 | grep "identical\|passed\|failed\|rror\|differs" | sed 's/identical to the oracle.*; k_wide/k_wide/' | tee -a $out/fullrun_t${t}_w$w.log
done; done
