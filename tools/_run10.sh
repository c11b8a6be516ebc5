F='Wcomment\|^ *[0-9]* |\|^ *|\|warning generated\|In file included\|amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_conv or cfg2_small or ns24 or cfg1_full or deterministic or opts_alt or clean_pair" 2>&1 | tail -3
for v in valu mfma; do
  if [ $v = mfma ]; then export DDP_G_MFMA=1; else unset DDP_G_MFMA; fi
  timeout 200 python tools/per_launch.py 2>&1 | grep -v "$F" | grep "conv32\|total" | awk '{printf "%s ", $2} END {print " <- '$v'"}'
done
unset DDP_G_MFMA
mkdir -p gpurun_out/r2h
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err
python -c "
import json
d=json.load(open('gpurun_out/r2h/bench.json')); r=d['roofline']
print(round(d['value'],2), round(d['ms_per_step'],2), r['kernel'], round(r['avg_launch_ms'],3), round(r['frac'],3), [(o['kernel'][4:14], round(o['avg_launch_ms'],3), o.get('ms_per_step')) for o in r['other_kernels']])"
timeout 300 python tools/stamp_conv.py > gpurun_out/r2h/stamps.log 2>&1; grep -v "$F" gpurun_out/r2h/stamps.log | sed -n 4,15p; grep "g_stage wave 0" gpurun_out/r2h/stamps.log
