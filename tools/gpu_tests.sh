#!/bin/bash
# usage: tools/gpu_tests.sh <outname> [pytest args...]  — the GPU suite (or a subset) with a log under gpurun_out/
out=gpurun_out/$1.log; shift
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu "$@" > $out 2>&1
echo "rc=$?" >> $out
grep -v amdgpu.ids $out | tail -n 25
