# round 6: stream priority of the direct conv's side stream (same-box bench pairs, captured steps), the range-flag test, smoke
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j6; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
python - <<PY
import torch
print("stream priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
PY
for p in "" 1 "" -1 ; do
  DDP_DIRECT_PRIO=$p timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass > $O/bench_p$p.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_p$p.json").read().strip().splitlines()[-1])
print("direct conv stream priority '$p':", round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step")
PY
done
timeout 600 python -m pytest tests -m gpu -q -k "range_flag or stage_a_plane or occupancy" > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
