# round 6, late: a direct conv as several tasks of segment ranges (model.direct_rows_max_split): parity subset, bench pairs at 40 and at 5 samples
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j24; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x -k "forward_matches_oracle or every_conv_output or single_conv" > $O/pytest_fwd.log 2>&1; tail -3 $O/pytest_fwd.log
for f in 1 6 1 6; do
  DDP_DIRECT_SPLIT=$f timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/bench40_s$f.json 2>> $O/bench.err
  DDP_DIRECT_SPLIT=$f timeout 600 python bench.py --samples 5 --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/bench5_s$f.json 2>> $O/bench.err
  python - <<PY
import json
for n in ("40","5"):
    d=json.loads(open("$O/bench%s_s$f.json" % n).read().strip().splitlines()[-1])
    r=d["roofline"]
    ks={k["kernel"]:round(k["avg_launch_ms"],3) for k in [r]+r["other_kernels"]}
    print("split=$f samples", n, round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step", ks)
PY
done
