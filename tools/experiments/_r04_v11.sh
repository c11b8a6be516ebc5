R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v11; mkdir -p $O; cd $R
B="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads > $O/bench.json 2>$O/bench.err; grep -o '"ms_per_step": [0-9.]*' $O/bench.json | head -1
DDP_REDUCE_NARROW=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads > $O/bench_narrow.json 2>$O/bench.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_narrow.json | head -1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B > $O/prof.log 2>&1; echo "prof rc=$?"
cd $R
python3 tools/step_sequence.py $O/prof > $O/prof.sequence.txt 2>&1; tail -1 $O/prof.sequence.txt
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*_agent_info.csv" -delete
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 $f
