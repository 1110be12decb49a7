#!/bin/bash
# usage: mk.sh name [extra -D flags]   (experiment build; pass -DCNS_ONLY_NPL=<w> to compile one tile width only)
n=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I../include \
  "$@" ../cranesched_amd/csrc/engine.hip -o v_$n.so 2> v_$n.log && echo built $n
