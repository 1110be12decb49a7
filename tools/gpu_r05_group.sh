export TMPDIR=/tmp
mkdir -p gpurun_out/r05e
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_host_adapter.py tests/test_abi.py -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/r05e/group_tests.log
cranesched_amd/host/test_host_adapter --group-check 65536 8 1000000 0,0 2>&1 | tail -5 | tee gpurun_out/r05e/group_check_full.txt
cranesched_amd/host/test_host_adapter --e2e-bench 65536 8 1000000 deferred 4 --devices 0,0 2>&1 | tail -7 | tee gpurun_out/r05e/e2e_two_engines.txt
