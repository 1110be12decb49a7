#!/bin/bash
# usage: tools/gpu_pmc.sh <outdir> <lib> <config>  — SQ counters of the selection kernel (separate passes, kernel-trace only)
out=gpurun_out/$1; lib=$2; cfg=$3
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  n=$1; shift
  CNS_ENGINE_LIB=$R/$lib timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $R/$out/$n -o $n --output-format csv -- python $R/tools/var_bench.py $cfg > $R/$out/$n.log 2>&1
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS
run p3 SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM
cd $R
python3 - <<PY
import csv, glob, collections
for n in ("p1","p2","p3"):
    fs = glob.glob("$out/%s/**/*counter_collection.csv" % n, recursive=True)
    if not fs: print(n, "no csv"); continue
    agg = collections.defaultdict(float)
    for row in csv.DictReader(open(fs[0])):
        k = row.get("Kernel_Name","")
        if "k_pipe" in k or "k_select" in k or "k_wide" in k:
            agg[(k[:40], row["Counter_Name"])] += float(row["Counter_Value"])
    for (k,c),v in sorted(agg.items()): print(n, k, c, v)
PY
