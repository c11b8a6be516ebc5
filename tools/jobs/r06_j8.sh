# round 6: stage A's drain with the hi and lo store of a group as consecutive instructions (-DDDP_SA_PAIR=1) against the product build
# (one k-step apart): stage A standalone and same-box bench pairs; the fixed range-flag test
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j8; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
python -c "from diffdock_pocket_amd import build; build.build(defs=['DDP_SA_PAIR=1'], tag='sapair', verbose=False)" >> $O/build.log 2>&1; echo "variant rc=$?"
V=$R/diffdock_pocket_amd/libddp_hip_sapair.so
for i in 1 2; do
  timeout 300 python tools/bench_stage_a.py 2>&1 | grep "plane form (ddp_stage_a_gh)" | sed 's/^/product  /'
  DDP_HIP_LIB=$V timeout 300 python tools/bench_stage_a.py 2>&1 | grep "plane form (ddp_stage_a_gh)" | sed 's/^/paired   /'
done
for v in "" $V "" $V; do
  DDP_HIP_LIB=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/b.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("lib '$v'"[-28:], round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step", [ (k["kernel"], round(k["avg_launch_ms"],3), round(k.get("ms_per_step",0),3)) for k in r["other_kernels"] if "stage_a" in k["kernel"]])
PY
done
DDP_HIP_LIB=$V timeout 600 python -m pytest tests -m gpu -q -k "stage_a or forward_matches_oracle_and_golden" > $O/pytest_variant.log 2>&1; tail -3 $O/pytest_variant.log
timeout 600 python -m pytest tests -m gpu -q -k "range_flag or split_products" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
