export TMPDIR=/tmp
bash tools/gpu_tests.sh r05_gpu_tests_a --durations=12
bash tools/gpu_bench.sh r05_bench_a
for c in C5 C2; do timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --config $c --no-cpu-baseline > gpurun_out/r05_bench_a/bench_$c.json 2> gpurun_out/r05_bench_a/bench_$c.err; tail -c 1500 gpurun_out/r05_bench_a/bench_$c.json; done
