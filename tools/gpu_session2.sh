#!/bin/bash
mkdir -p gpurun_out/s2
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/s2/suite.log 2>&1; echo "suite rc=$?" >> gpurun_out/s2/suite.log
CNS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/s2/bench_dist.json 2> gpurun_out/s2/bench_dist.err; echo "dist rc=$?"
./cranesched_amd/host/test_host_adapter --cycle-bench 16384 1000000 > gpurun_out/s2/cycle_bench.txt 2>&1
./cranesched_amd/host/test_host_adapter --mirror-check 65536 300000 > gpurun_out/s2/mirror_check.txt 2>&1
grep -v amdgpu.ids gpurun_out/s2/suite.log | tail -n 6; python -c "
import json; d=json.load(open('gpurun_out/s2/bench_dist.json')); print({k:d[k] for k in ('value','ms_per_step','allgather_merged_identical_to_single_gpu') if k in d})"
tail -n 8 gpurun_out/s2/cycle_bench.txt; tail -n 7 gpurun_out/s2/mirror_check.txt
