#!/bin/bash
# which kernel of a build with a forced `s_waitcnt 0` before every instruction faults (a pure timing change: whatever breaks under it is a latent ordering bug)
export TMPDIR=/tmp CNS_WIDE_NO_RETRY=1
for k in legacy pipe wide; do for c in C1 C2; do
  echo "== $k $c"; CNS_SELECT_KERNEL=$k CNS_ENGINE_LIB=build_var/v_fz.so timeout 120 python tools/var_bench.py $c 2>&1 | grep -v amdgpu | tail -2
done; done
