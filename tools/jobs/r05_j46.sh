R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j46; mkdir -p $O; cd $R
ulimit -c 0
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['metric'], round(d['value'],2), d['unit'], d['ms_per_step'], d['n_gpus'], d['steps'], d['warmup']); print(d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['pmc_source'].get('stale')); print(d['cpu_baseline']['value'])"
