# A/B of engine builds: tools/r06/ab.sh <outdir> "<configs>" lib[:AUX] ...
out=gpurun_out/$1; mkdir -p $out; cfgs=$2; shift 2
export CNS_WIDE_NO_RETRY=1
for v in "$@"; do
  lib=${v%%:*}; aux=${v#*:}; [ "$aux" = "$v" ] && aux=""
  env ${aux:+CNS_WIDE_AUX=$aux} CNS_ENGINE_LIB=build_var/v_$lib.so python tools/var_bench.py $cfgs 2>&1 | sed "s/^/aux=$aux /" | tee -a $out/ab.txt
done
