R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j11; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ulimit -c 0
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass --no-overlap-direct"
for v in "" _ghabl1 _ghct4 _ghct2; do
export DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip$v.so
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$v -- $B > $O/prof$v.log 2>&1
python3 - <<PY
import glob, pandas as pd, os
f=max(glob.glob("$O/prof$v/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
d=pd.read_csv(f); d["Name"]=d["Name"].str.slice(0,40)
print("lib$v", d[d.Name.str.contains("stage_a")][["Name","Calls","AverageNs","MaxNs"]].to_string(header=False))
PY
done
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*_agent_info.csv" -delete
