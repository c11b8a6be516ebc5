# round 6, late: whole-tile G ring (GK = 6) for the scalar segments now that a fragment's lo half is two registers (-DR16_GK_FULL=1): bench pairs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j21; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
python -c "from diffdock_pocket_amd import build; print(build.build(defs=['R16_GK_FULL=1'], tag='gkfull'))" >> $O/build.log 2>&1; echo "variant rc=$?"
for f in base gkfull base gkfull; do
  if [ $f = base ]; then unset DDP_HIP_LIB; else export DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_gkfull.so; fi
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/bench_$f.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$f", round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step; rows launch", round(r["avg_launch_ms"],3), "ms; frac", round(r["frac"],4), {k:round(v["avg_launch_ms"],3) for k,v in r["by_layer"].items()})
PY
done
unset DDP_HIP_LIB
DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_gkfull.so timeout 600 python -m pytest tests -m gpu -q -x -k "forward_matches_oracle" > $O/pytest_fwd.log 2>&1; tail -2 $O/pytest_fwd.log
