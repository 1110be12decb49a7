#!/bin/bash
# usage: tools/gpu_widths.sh <out> "<configs>" "<kernel settings>" — one k_wide run per configuration and CNS_SELECT_KERNEL setting (retry off)
export TMPDIR=/tmp
out=gpurun_out/$1; : > $out
for cfg in $2; do for k in $3; do
  echo "== $cfg $k" >> $out
  CNS_WIDE_NO_RETRY=1 CNS_SELECT_KERNEL=$k timeout 200 python tools/prof_wide.py $cfg 2>&1 | grep "us/job\|rror" | cut -c1-120 >> $out
done; done
cat $out
