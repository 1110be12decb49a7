#!/bin/bash
# experiment builds of the engine: build_var/mk.sh <tag> <extra hipcc flags...>  ->  build_var/v_<tag>.so
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing -fPIC -shared "$@" cranesched_amd/csrc/engine.hip -o build_var/v_$tag.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -lpthread 2> build_var/$tag.log || { tail -20 build_var/$tag.log; exit 1; }
echo built v_$tag.so
