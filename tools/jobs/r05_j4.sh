R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j4; mkdir -p $O; cd $R
ulimit -c 0
timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids > $O/stamps_l3.txt; cat $O/stamps_l3.txt
STAMP_LAYER=1 timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids > $O/stamps_l1.txt; cat $O/stamps_l1.txt
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
timeout 300 $B > $O/bench.json 2>$O/bench.err; echo "pipelined: $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json | head -1)"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B --no-overlap-direct > $O/prof.log 2>&1
python3 - <<PY
import glob, pandas as pd, os
f=max(glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
d=pd.read_csv(f); d["Name"]=d["Name"].str.slice(0,60)
print(d.head(5)[["Name","Calls","AverageNs","MinNs","MaxNs","Percentage"]].to_string())
PY
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*_agent_info.csv" -delete
