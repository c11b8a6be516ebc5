# round 6, late: the layers' direct convs through the row-stationary kernel (model.direct_rows): parity subset, bench pairs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j20; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x -k "forward_matches_oracle or every_conv_output" > $O/pytest_fwd.log 2>&1; tail -6 $O/pytest_fwd.log
for f in 0 1 0 1; do
  DDP_DIRECT_ROWS=$f timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/bench_d$f.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_d$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("direct_rows=$f", round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step; rows launch", round(r["avg_launch_ms"],3), "ms; frac", round(r["frac"],4), [ (k["kernel"], round(k["avg_launch_ms"],3), round(k.get("ms_per_step",0),3)) for k in r["other_kernels"]])
PY
done
tail -3 $O/bench.err
