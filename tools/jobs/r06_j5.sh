# round 6: plane form 2 against form 0 (same-box bench pairs + stage A standalone), the pair's start / end offsets under rocprofv3, the changed tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j5; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
for f in 0 2 0 2; do
  DDP_GH_FMT=$f timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/bench_f$f.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_f$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("gh_fmt=$f", round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step; rows launch", round(r["avg_launch_ms"],3), "ms; frac", round(r["frac"],4), [ (k["kernel"], round(k["avg_launch_ms"],3), round(k.get("ms_per_step",0),3)) for k in r["other_kernels"]])
PY
done
DDP_GH_FMT=2 timeout 900 python -m pytest tests -m gpu -q -x -k "forward_matches_oracle or single_conv or every_conv_output or stage_a_plane or range_flag or occupancy" > $O/pytest_f2.log 2>&1; tail -4 $O/pytest_f2.log
timeout 600 python -m pytest tests -m gpu -q -k "stage_a_plane or range_flag or sampler_end_to_end or cfg2_job_end_to_end" --durations=8 > $O/pytest_new.log 2>&1; tail -14 $O/pytest_new.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/pair_trace -- python3 $R/tools/overlap_ab.py --pair-only > $O/pair_trace.log 2>&1; echo "pair trace rc=$?"
cd $R
python3 tools/pair_offsets.py $O/pair_trace > $O/pair_offsets.txt 2>&1; cat $O/pair_offsets.txt
find $O/pair_trace -name "*.csv" -size +1M -delete
