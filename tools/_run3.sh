ulimit -c 0
O=gpurun_out/r03_h4; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_lists.py tests/test_gpu_parity.py -m gpu -q -x -k "lists or scan or group or node_encoders or forward_matches or capacities or graph_replay or static_graph or weight_edits or pyg_shaped or device_driven" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- $B > $R/$O/prof.log 2>&1; echo "prof rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_5 -- $B --samples 5 > $R/$O/prof_5.log 2>&1; echo "prof 5 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cfg1 -- $B --samples 4 --cfg cfg1 --flex > $R/$O/prof_cfg1.log 2>&1; echo "prof cfg1 rc=$?"
cd $R
for d in prof prof_5 prof_cfg1; do python3 tools/step_sequence.py $O/$d > $O/$d.sequence.txt 2>&1; tail -1 $O/$d.sequence.txt; grep -h "ms_per_step" $O/$d.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
for a in "" "--samples 5" "--samples 4 --cfg cfg1 --flex"; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads $a 2>>$O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['ms_per_step'], d['value'])"; done
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*_agent_info.csv" -delete
