R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j19; mkdir -p $O; cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "single_conv_layer or test_every_conv_output or stage_a or test_forward_matches" 2>&1 | tail -3
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for i in 1 2; do timeout 300 $B > $O/b.json 2>$O/b.err; echo "rigid rows: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"; done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B --no-overlap-direct > $O/prof.log 2>&1
python3 - <<PY
import glob, pandas as pd, os
f=max(glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
d=pd.read_csv(f); d["Name"]=d["Name"].str.slice(0,60)
print(d.head(4)[["Name","Calls","AverageNs","MinNs","MaxNs","Percentage"]].to_string())
PY
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*_agent_info.csv" -delete
