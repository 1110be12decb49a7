#!/bin/bash
# usage: tools/gpu_narrow.sh <tag>  — the many-partition builds of k_wide: their test module, then C4p64 under the default choice and under k_pipe
export TMPDIR=/tmp
tag=$1
timeout 600 python -m pytest tests/test_gpu_wide_narrow.py -q -m gpu -x --durations=8 > gpurun_out/${tag}_tests.log 2>&1
echo "rc=$?" >> gpurun_out/${tag}_tests.log
grep -v amdgpu.ids gpurun_out/${tag}_tests.log | tail -n 22
timeout 200 python bench.py --config C4p64 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_c4p64.json 2> gpurun_out/${tag}_bench_c4p64.err
echo "bench rc=$?"; cat gpurun_out/${tag}_bench_c4p64.json | head -c 1500; echo
CNS_SELECT_KERNEL=pipe timeout 200 python bench.py --config C4p64 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_c4p64_pipe.json 2> gpurun_out/${tag}_bench_c4p64_pipe.err
echo "bench pipe rc=$?"; cat gpurun_out/${tag}_bench_c4p64_pipe.json | head -c 600; echo
