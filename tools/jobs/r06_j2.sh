# round 6: the 3-byte G plane (fp16 hi + e4m3 lo): unit test of both stage-A plane forms, the GPU suite under DDP_G_PLANES3=1, stage A standalone,
# same-box bench pairs of the two forms
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j2; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"   # (the snapshot may hold sources edited after the last local build)
timeout 300 python -m pytest tests -m gpu -x -q -k "stage_a_plane_forms" > $O/pytest_planes.log 2>&1; tail -5 $O/pytest_planes.log
timeout 300 python tools/bench_stage_a.py > $O/stage_a.txt 2>&1; grep "plane form" $O/stage_a.txt
for f in 0 1 0 1; do
  DDP_G_PLANES3=$f timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/bench_g$f.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_g$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("g_planes3=$f", round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step; rows launch", round(r["avg_launch_ms"],3), "ms; frac", round(r["frac"],4), [ (k["kernel"], round(k["avg_launch_ms"],3), round(k.get("ms_per_step",0),3)) for k in r["other_kernels"]])
PY
done
DDP_G_PLANES3=1 timeout 2400 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_g3.log 2>&1; tail -15 $O/pytest_g3.log
