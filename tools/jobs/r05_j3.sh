# round 5, job 3: serial launch order (no overlap), rows vs 32-edge kernel, rocprof kernel stats of both
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ulimit -c 0
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass --no-overlap-direct"
for v in 1 0; do
export DDP_CONV_ROWS=$v
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$v -- $B > $O/prof$v.log 2>&1; echo "prof rows=$v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/prof$v.log | head -1)"
python3 - <<PY
import glob, pandas as pd, os
f=max(glob.glob("$O/prof$v/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
d=pd.read_csv(f)
d["Name"]=d["Name"].str.slice(0,60)
print(d.head(6)[["Name","Calls","AverageNs","MinNs","MaxNs","Percentage"]].to_string())
PY
done
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*_agent_info.csv" -delete
