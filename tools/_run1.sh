set -x
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -15 gpurun_out/r2a/pytest.log
timeout 300 python bench.py --steps 20 --warmup 2 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc=$?"
timeout 200 env DDP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 1 --no-cpu-baseline --scaling strong > gpurun_out/r2a/bench_2rank_strong.json 2> gpurun_out/r2a/bench_2rank.err; echo "2rank rc=$?"
timeout 200 python bench.py --samples 5 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_5samples.json 2> gpurun_out/r2a/bench_5.err; echo "5samples rc=$?"
timeout 200 python tools/per_launch.py > gpurun_out/r2a/per_launch.log 2>&1
timeout 300 python tools/stamp_conv.py > gpurun_out/r2a/stamps.log 2>&1
tail -5 gpurun_out/r2a/bench.err gpurun_out/r2a/bench_2rank.err gpurun_out/r2a/bench_5.err
cat gpurun_out/r2a/per_launch.log
