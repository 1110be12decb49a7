out=gpurun_out/${1:-r06b}; mkdir -p $out
export CNS_WIDE_NO_RETRY=1
for a in ${AUXS:-0 1 2 3}; do for c in ${CFGS:-C5 C4 C2}; do
  CNS_WIDE_AUX=$a timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $out/aux${a}_$c.json 2> $out/aux${a}_$c.err
done; done
CNS_SELECT_KERNEL=wide timeout 900 python -m pytest tests/test_gpu_fullrun.py -x -q -k "wide and (${TESTS:-c5 or c4 or c2 or tile1})" -s > $out/fullrun_wide.log 2>&1
tail -3 $out/fullrun_wide.log
