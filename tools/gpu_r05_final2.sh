#!/bin/bash
# round 5, second record run (after the host pass of cns_upload_jobs moved onto host threads): the bracket of SURVEY 8(d) first (bench line
# without the CPU legs), the adapter's end-to-end cycle at 16 / 32 host threads, then the whole GPU suite.  Everything under its own timeout;
# results appear under gpurun_out/<tag>/ as they finish.
export TMPDIR=/tmp
t=${1:-r05_final2}
out=gpurun_out/$t; mkdir -p $out
nproc > $out/host.txt; lscpu | grep "Model name" >> $out/host.txt
timeout 240 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err
echo "bench rc=$?"; python3 - <<PY
import json
try:
    d = json.load(open("$out/bench_n1.json"))
    print("value", d["value"], "ms_per_step", d["ms_per_step"], "incl", json.dumps(d.get("incl_h2d_d2h"))[:900])
except Exception as e:
    print("bench line unreadable:", e); print(open("$out/bench_n1.err").read()[-2000:])
PY
for th in 16 32; do echo "== $th host threads"; timeout 120 cranesched_amd/host/test_host_adapter --e2e-bench 65536 8 1000000 deferred $th 2>&1 | grep "cycle [123]\|ok\|FAIL\|error" ; done | tee $out/adapter_e2e.txt
CNS_HOST_THREADS=1 timeout 120 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_n1_one_host_thread.json 2>/dev/null
python3 -c "import json; d=json.load(open('$out/bench_n1_one_host_thread.json')); print('one host thread: incl', json.dumps(d.get('incl_h2d_d2h'))[:700])"
timeout 900 python -m pytest tests -q -m gpu --durations=8 > $out/gpu_tests.log 2>&1
echo "rc=$?" >> $out/gpu_tests.log
grep -v amdgpu.ids $out/gpu_tests.log | tail -n 25
# (if the box still has time) the bracket of the other BASELINE configurations under this build
for c in C5 C2; do timeout 200 python bench.py --gpus 1 --steps 3 --warmup 1 --config $c --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err; python3 -c "import json; d=json.load(open('$out/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], json.dumps(d.get('incl_h2d_d2h'))[:400])"; done
