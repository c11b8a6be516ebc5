# round 5, job 2: first timing of ddp_conv_rows against the 32-edge kernel, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j2; mkdir -p $O; cd $R
ulimit -c 0
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads"
timeout 300 $B > $O/bench_rows.json 2> $O/bench_rows.err; echo "rows: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_rows.json | head -3 | tr '\n' ' ')"; tail -3 $O/bench_rows.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B --no-roofline-pass > $O/prof.log 2>&1; echo "prof rc=$?"
cd $R
python3 - <<'PY'
import glob, pandas as pd, os
f=max(glob.glob(os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r05_j2/prof/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
d=pd.read_csv(f)
print(d.head(14).to_string())
PY
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*_agent_info.csv" -delete
