# round 6: rows launches on a HIGH-priority stream (the direct conv then fills what they leave), same-box pairs; the range-flag test again
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j7; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
for p in "" -1 "" -1 ; do
  DDP_ROWS_PRIO=$p timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass > $O/bench_p$p.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_p$p.json").read().strip().splitlines()[-1])
print("rows stream priority '$p':", round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step")
PY
done
DDP_ROWS_PRIO=-1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass --flex > $O/bench_flex_p.json 2>> $O/bench.err; python -c "
import json;d=json.loads(open('$O/bench_flex_p.json').read().strip().splitlines()[-1]);print('flex, rows priority -1:',round(d['ms_per_step'],3))"
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass --flex > $O/bench_flex.json 2>> $O/bench.err; python -c "
import json;d=json.loads(open('$O/bench_flex.json').read().strip().splitlines()[-1]);print('flex, default:',round(d['ms_per_step'],3))"
DDP_ROWS_PRIO=-1 timeout 600 python -m pytest tests -m gpu -q -k "pipelined_layer_order or range_flag or split_products" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
