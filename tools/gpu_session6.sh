#!/bin/bash
# full -m gpu suite (both selection kernels) + the default bench line
mkdir -p gpurun_out/s6
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/s6/suite.log 2>&1; echo "suite rc=$?" >> gpurun_out/s6/suite.log
timeout 600 python bench.py > gpurun_out/s6/bench.json 2> gpurun_out/s6/bench.err; echo "bench rc=$?"
grep -v amdgpu.ids gpurun_out/s6/suite.log | tail -n 6; cat gpurun_out/s6/bench.json
