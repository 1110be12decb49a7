# The round's record run on one GPU box: tools/r06/record.sh <outdir>
#   bench line + rocprofv3 kernel stats + HBM PMC passes (tools/gpu_bench.sh), the other configs' bench lines, the C++ adapter's
#   end-to-end cycle, a short soak of the multi-home protocol, the whole GPU suite.
o=${1:-r06rec2}; out=gpurun_out/$o; mkdir -p $out
bash tools/gpu_bench.sh $o > $out/script.log 2>&1
for c in C5 c5deep C2 C3 C4r C4p64 C4p256; do
  timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err
done
( for t in 32 16; do timeout 300 cranesched_amd/host/test_host_adapter --e2e-bench 65536 8 1000000 deferred $t; done ) > $out/adapter_e2e.txt 2>&1
timeout 900 python tools/r06/soak.py ${SOAK:-12} c4 c4v c5 2>&1 | grep -v amdgpu.ids > $out/soak.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=25 2>&1 | grep -v amdgpu.ids > $out/gpu_tests.log
tail -3 $out/gpu_tests.log; tail -3 $out/soak.txt; grep -h "cycle 3" $out/adapter_e2e.txt
for c in n1 C5 c5deep C2 C3 C4r C4p64 C4p256; do python - <<PY
import json
try:
    d = json.loads(open("$out/bench_$c.json").read().strip().splitlines()[-1])
    print("$c", d["value"], d["unit"], d["ms_per_step"], "ms", d.get("roofline", {}).get("frac"))
except Exception as e:
    print("$c", "unreadable:", e)
PY
done
