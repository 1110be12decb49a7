#!/bin/bash
# usage: tools/gpu_bench.sh <outdir>  — the driver's bench line + rocprofv3 kernel stats + HBM PMC passes of the same command
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 > $out/bench_n1.json 2> $out/bench_n1.err
echo "bench rc=$?"; tail -c 3000 $out/bench_n1.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/stats -o stats --output-format csv -- python $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $R/$out/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$out/pmc_fetch -o f --output-format csv -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $R/$out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$out/pmc_write -o w --output-format csv -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $R/$out/pmc_write.log 2>&1
cd $R
find $out -name "*kernel_stats.csv" | head -2 | xargs -I{} head -12 {}
python3 - <<PY
import csv, glob, collections
for n, c in (("pmc_fetch","FETCH_SIZE"),("pmc_write","WRITE_SIZE")):
    fs = glob.glob("$out/%s/**/*counter_collection.csv" % n, recursive=True)
    if not fs: print(n, "no csv"); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(fs[0])):
        k = row.get("Kernel_Name","")
        if row["Counter_Name"] == c:
            agg[k[:48]][0] += float(row["Counter_Value"]); agg[k[:48]][1] += 1
    for k,(v,nl) in sorted(agg.items(), key=lambda x:-x[1][0])[:6]: print(n, k, c, "total", v, "launches", nl, "per launch", v/nl)
PY
