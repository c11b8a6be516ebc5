cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in 5 40; do
O=$R/gpurun_out/r03_p1_$S
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --samples $S --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass > $O/log.txt 2>&1
echo "rc=$?"
tail -1 $O/log.txt | cut -c1-300
f=$(ls $O/*/*kernel_stats.csv | head -1)
python3 - <<PY
import pandas as pd
t=pd.read_csv("$f")
t=t.sort_values("TotalDurationNs",ascending=False)
print(t[["Name","Calls","TotalDurationNs","AverageNs","Percentage"]].head(32).to_string())
print("total ms/step (23 steps):", t.TotalDurationNs.sum()/23/1e6, "launches/step:", t.Calls.sum()/23)
PY
python3 $R/tools/gaps.py $O | head -14
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*_agent_info.csv" -delete
done
