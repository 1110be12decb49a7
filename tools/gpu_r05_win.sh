#!/bin/bash
# round 5, a window of jobs per pool exchange: small-case parity under k_wide (64 waves), the full-size digests with the window on / off,
# and the window path by segment (build_var/v_pwin.so = -DCNS_PROF_WIN)
export TMPDIR=/tmp
out=gpurun_out/${1:-r05b}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_reservations.py -q -m gpu -x -k "wide and not wide32" > $out/parity.log 2>&1; tail -3 $out/parity.log
for w in ${WINS:-16 0 8}; do
  echo "== CNS_WIDE_WINDOW=$w"
  CNS_WIDE_WINDOW=$w timeout 600 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide and not wide32 and (c2 or c4] or c5 or c4r])" 2>&1 | grep "identical\|passed\|failed\|rror\|differs" | sed 's/identical to the oracle.*; k_wide/k_wide/' | tee -a $out/fullrun_w$w.log
done
if [ -f build_var/v_pwin.so ]; then for c in C5 C4 C2; do CNS_ENGINE_LIB=build_var/v_pwin.so python tools/prof_win.py $c 2>&1 | grep -v amdgpu.ids; done | tee $out/prof_win.txt; fi
