# round 6, late: stage A (plane form 1) with free-running waves (no shared x tile, no barrier): unit tests, stage A alone, bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j19; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
timeout 300 python -m pytest tests -m gpu -q -k "stage_a_plane_forms" > $O/pytest_planes.log 2>&1; tail -3 $O/pytest_planes.log
timeout 300 python tools/bench_stage_a.py > $O/stage_a.txt 2>&1; grep "plane form" $O/stage_a.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "forward_matches_oracle" > $O/pytest_fwd.log 2>&1; tail -3 $O/pytest_fwd.log
for f in 1 1; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/bench_$f.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step; rows launch", round(r["avg_launch_ms"],3), "ms; frac", round(r["frac"],4), [ (k["kernel"], round(k["avg_launch_ms"],3), round(k.get("ms_per_step",0),3)) for k in r["other_kernels"]])
PY
done
