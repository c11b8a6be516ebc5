R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v9; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x -k "stage_a or fp16_split or work_eliminations" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_sel.log
timeout 300 python tools/bench_stage_a.py 2>&1 | grep -v amdgpu.ids > $O/stage_a.txt; cat $O/stage_a.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
timeout 900 $B > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_v9/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['achieved'], r['peak'], r['fp32_equivalent_tflops'])
print({t:round(x['avg_launch_ms'],2) for t,x in r['by_layer'].items()})
for o in r['other_kernels']: print(o['kernel'], round(o.get('ms_per_step',0),2), o.get('frac'))
for k,v in d['other_workloads'].items(): print(k, round(v['value'],1), round(v['ms_per_step'],2))
PY
