R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j12; mkdir -p $O; cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "single_conv_layer and 60-10 or test_every_conv_output or test_forward_matches_oracle_and_golden or stage_a or fp16_range or fp16_split" 2>&1 | tail -5
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for i in 1 2; do
timeout 300 $B > $O/bench.json 2>$O/bench.err; echo "rows: $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json | head -1)"
DDP_CONV_ROWS=0 timeout 300 $B > $O/bench0.json 2>$O/bench0.err; echo "conv32: $(grep -o '"ms_per_step": [0-9.]*' $O/bench0.json | head -1)"
done
timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids > $O/stamps_l3.txt; head -3 $O/stamps_l3.txt; tail -1 $O/stamps_l3.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B --no-overlap-direct > $O/prof.log 2>&1
python3 - <<PY
import glob, pandas as pd, os
f=max(glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
d=pd.read_csv(f); d["Name"]=d["Name"].str.slice(0,60)
print(d.head(5)[["Name","Calls","AverageNs","MinNs","MaxNs","Percentage"]].to_string())
PY
cd $R
DDP_TRAJ_LOG=$O/cfg2_traj.txt timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "test_cfg2_job_end_to_end" 2>&1 | tail -5; cat $O/cfg2_traj.txt
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*_agent_info.csv" -delete
