# round 6: rows16 with the two vector blocks' G runs as one pass (product build of this tree) against separate passes (-DR16_PAIR=0): same-box
# bench pairs; parity tests under the pair build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j12; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
python -c "from diffdock_pocket_amd import build; build.build(defs=['R16_PAIR=0'], tag='r16nopair', verbose=False)" >> $O/build.log 2>&1; echo "variant rc=$?"
V=$R/diffdock_pocket_amd/libddp_hip_r16nopair.so
line() { python - "$1" "$2" <<PY
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step; rows launch", round(r["avg_launch_ms"],3), "ms; frac", round(r["frac"],4), {k:round(v["avg_launch_ms"],3) for k,v in r["by_layer"].items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for i in 1 2 3; do
  DDP_HIP_LIB=$V timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/b.json 2>> $O/bench.err; line $O/b.json "separate passes "
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/b.json 2>> $O/bench.err; line $O/b.json "pair            "
done
timeout 1500 python -m pytest tests -m gpu -q -k "single_conv or forward_matches_oracle or every_conv_output or operand_planes or capacities or sharing or pruning" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
