F='Wcomment\|^ *[0-9]* |\|^ *|\|warning generated\|In file included\|amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_conv or cfg2_small or ns24 or cfg1_full or deterministic or clean_pair" 2>&1 | tail -2
timeout 200 python tools/per_launch.py 2>&1 | grep -v "$F" | grep "conv32\|total" | awk '{printf "%s ", $2} END {print " <- unrolled tile loop"}'
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-other-workloads | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('40 samples', round(d['value'],2), round(d['ms_per_step'],2), round(r['avg_launch_ms'],3), round(r['frac'],3))"
timeout 300 python tools/stamp_conv.py 2>&1 | grep -v "$F" | sed -n 4,15p
