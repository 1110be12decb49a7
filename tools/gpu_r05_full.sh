#!/bin/bash
# round 5, the record run: the whole GPU suite, the driver's bench line + rocprofv3 kernel stats + HBM PMC passes, the other BASELINE
# configurations' bench lines, the adapter's end-to-end cycle and the multi-engine check on one GPU
export TMPDIR=/tmp
t=${1:-r05_final}
bash tools/gpu_tests.sh ${t}_gpu_tests --durations=8
bash tools/gpu_bench.sh ${t}_bench
for c in C5 C2 C3; do timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --config $c --no-cpu-baseline > gpurun_out/${t}_bench/bench_$c.json 2> gpurun_out/${t}_bench/bench_$c.err; head -c 330 gpurun_out/${t}_bench/bench_$c.json; echo; done
CNS_WIDE_WINDOW=16 timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --config C5 --no-cpu-baseline > gpurun_out/${t}_bench/bench_C5_windows.json 2>/dev/null; head -c 330 gpurun_out/${t}_bench/bench_C5_windows.json; echo
cranesched_amd/host/test_host_adapter --e2e-bench 65536 8 1000000 deferred 4 2>&1 | tail -5 | tee gpurun_out/${t}_bench/adapter_e2e.txt
cranesched_amd/host/test_host_adapter --group-check 65536 8 1000000 0,0 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -4 | tee gpurun_out/${t}_bench/group_check.txt
