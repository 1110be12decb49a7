#!/bin/bash
# the adapter's end-to-end cycle (1 M jobs, 8 partitions, deferred write-back) by number of host threads
export TMPDIR=/tmp
nproc
for t in 8 16 32 64; do echo "== $t host threads"; cranesched_amd/host/test_host_adapter --e2e-bench 65536 8 1000000 deferred $t 2>&1 | grep "cycle [123]" ; done
