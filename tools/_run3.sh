ulimit -c 0
O=gpurun_out/r03_h7; mkdir -p $O
for a in "" "--stage-a-fp32" "" "--stage-a-fp32" "--samples 5" "--samples 5 --stage-a-fp32"; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads $a 2>>$O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$a |', round(d['ms_per_step'],3), '| conv32 avg launch', round(r['avg_launch_ms'],4), [ (e['kernel'][:18], round(e.get('ms_per_step') or 0,3)) for e in r.get('other_kernels',[])])"; done 2>&1 | tee $O/ab.txt
