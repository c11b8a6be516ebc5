# round 6, end: the direct convs' split at 2, 3, 6 ranges bit for bit against 1 (new test); the shard sizes of the strong split (20, 10 samples per rank) with and without it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j26; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests -m gpu -q -k "segment_ranges_change_no_bit or forward_direct_path" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for n in 20 10; do for f in 1 6 1 6; do
  DDP_DIRECT_SPLIT=$f timeout 600 python bench.py --samples $n --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/bench${n}_s$f.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench${n}_s$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
ks={k["kernel"]:round(k["avg_launch_ms"],3) for k in [r]+r["other_kernels"] if "conv" in k["kernel"]}
print("samples $n split=$f", round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step", ks)
PY
done; done
