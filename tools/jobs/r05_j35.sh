R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j35; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ulimit -c 0
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
for v in new rd16 rd4; do
  unset DDP_HIP_LIB
  [ $v != new ] && export DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_$v.so
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- $B > $O/prof_$v.log 2>&1
  f=$(ls $O/prof_$v/*/*kernel_stats.csv | head -1)
  echo "== $v: $(grep -o '"ms_per_step": [0-9.]*' $O/prof_$v.log | head -1)"; grep -E "segment_reduce" $f | cut -d, -f1-4
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete
done
