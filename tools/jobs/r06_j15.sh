# round 6: stage A's plane form at three waves per SIMD (-DDDP_SAH_W3: two column tiles per wave, one x buffer, 46 KB of LDS) against the product
# (two 253-register waves): standalone and same-box bench pairs; parity of the variant
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j15; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
python -c "from diffdock_pocket_amd import build; build.build(defs=['DDP_SAH_W3'], tag='saw3', verbose=False)" >> $O/build.log 2>&1; echo "variant rc=$?"
V=$R/diffdock_pocket_amd/libddp_hip_saw3.so
for i in 1 2; do
  timeout 300 python tools/bench_stage_a.py 2>&1 | grep "plane form (ddp_stage_a_gh)" | sed 's/^/product  /'
  DDP_HIP_LIB=$V timeout 300 python tools/bench_stage_a.py 2>&1 | grep "plane form (ddp_stage_a_gh)" | sed 's/^/3 waves  /'
done
for v in "" $V "" $V; do
  DDP_HIP_LIB=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/b.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("lib '$v'"[-24:], round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step", [ (k["kernel"], round(k["avg_launch_ms"],3), round(k.get("ms_per_step",0),3)) for k in r["other_kernels"] if "stage_a" in k["kernel"]])
PY
done
DDP_HIP_LIB=$V timeout 900 python -m pytest tests -m gpu -q -k "stage_a or forward_matches_oracle_and_golden or every_conv" > $O/pytest_variant.log 2>&1; tail -3 $O/pytest_variant.log
