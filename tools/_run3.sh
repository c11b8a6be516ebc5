ulimit -c 0
O=gpurun_out/r03_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], r.get('mfma_busy_pmc'), r.get('padding_frac_pmc'), d['cpu_baseline']['value'])"
