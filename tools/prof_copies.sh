cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_p2
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --samples 5 --steps 6 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass > $O/log.txt 2>&1
python3 $R/tools/step_sequence.py $O
find $O -name "*kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete
