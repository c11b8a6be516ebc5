set -x
mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_conv or forward_matches or direct_path or deterministic or pruning or layer0" > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -15 gpurun_out/r2b/pytest.log
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('gpurun_out/r2b/bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], [(o['kernel'], o['avg_launch_ms']) for o in r['other_kernels']])"
timeout 200 python tools/per_launch.py 2>&1 | grep -v Wcomment | tail -16
timeout 300 python tools/stamp_conv.py > gpurun_out/r2b/stamps.log 2>&1; grep -v "Wcomment\|^ *[0-9]* |\|^ *|\|warning generated\|In file included" gpurun_out/r2b/stamps.log | tail -40
