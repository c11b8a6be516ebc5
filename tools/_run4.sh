mkdir -p gpurun_out/r2d
F='Wcomment\|^ *[0-9]* |\|^ *|\|warning generated\|In file included\|amdgpu.ids'
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_conv or cfg2_small or deterministic" 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err
python -c "
import json
d=json.load(open('gpurun_out/r2d/bench.json')); r=d['roofline']
print(round(d['value'],2), round(d['ms_per_step'],2), r['kernel'], round(r['avg_launch_ms'],3), round(r['frac'],3), [(o['kernel'][4:14], round(o['avg_launch_ms'],3)) for o in r['other_kernels']])"
timeout 200 python tools/per_launch.py 2>&1 | grep -v "$F" | grep "conv32\|total"
timeout 300 python tools/stamp_conv.py > gpurun_out/r2d/stamps.log 2>&1; grep -v "$F" gpurun_out/r2d/stamps.log | head -40
