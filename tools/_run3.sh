mkdir -p gpurun_out/r2c
F='Wcomment\|^ *[0-9]* |\|^ *|\|warning generated\|In file included\|amdgpu.ids'
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_conv or cfg2_small or cfg1_full or deterministic or ns24" 2>&1 | tail -3
for v in x6 w4; do
  if [ $v = w4 ]; then export DDP_CONV32_WAVES4=1; else unset DDP_CONV32_WAVES4; fi
  timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/r2c/bench_$v.json 2> gpurun_out/r2c/bench_$v.err
  python -c "
import json
d=json.load(open('gpurun_out/r2c/bench_$v.json')); r=d['roofline']
print('$v', round(d['value'],2), round(d['ms_per_step'],2), r['kernel'], round(r['avg_launch_ms'],3), round(r['frac'],3), [(o['kernel'][4:14], round(o['avg_launch_ms'],3)) for o in r['other_kernels']])"
  timeout 200 python tools/per_launch.py 2>&1 | grep -v "$F" | grep "conv32\|total"
done
unset DDP_CONV32_WAVES4
timeout 300 python tools/stamp_conv.py > gpurun_out/r2c/stamps.log 2>&1; grep -v "$F" gpurun_out/r2c/stamps.log | head -24
