# round 5, job 0: what does ddp_conv32_kernel<60,h2> wait for?  PMC wave-state + L1/L2 passes, no-weight / no-G timing ablations,
# and the micro-benchmark of a row-stationary tile loop (weights once per 256 edges through an LDS ring).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j0; mkdir -p $O; cd $R
ulimit -c 0
( cd tools/micro && ./stream_tiles > $O/stream_tiles.txt 2>&1; echo "micro rc=$?" )
cat $O/stream_tiles.txt
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads"
for v in "" _ablate1 _ablate3 _ablate7; do
  DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip$v.so timeout 300 $B > $O/bench$v.json 2> $O/bench$v.err
  echo "lib$v: $(grep -o '"ms_per_step": [0-9.]*' $O/bench$v.json | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/bench$v.json | head -1)"
done
cd /tmp; export TMPDIR=/tmp
P="python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-hbm-pass --no-other-workloads --no-roofline-pass"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -- $P > $O/pmc_sq.log 2>&1; echo "sq rc=$?"
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_l1l2 -- $P > $O/pmc_l1l2.log 2>&1; echo "l1l2 rc=$?"
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_lat -- $P > $O/pmc_lat.log 2>&1; echo "lat rc=$?"
cd $R
for d in pmc_sq pmc_l1l2 pmc_lat; do python3 tools/pmc_raw.py $O/$d > $O/$d.json 2> $O/$d.err; echo "$d: $(wc -c < $O/$d.json) bytes"; done
cat $O/pmc_l1l2.json | head -20
find $O -name "*counter_collection.csv" -size +2M -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O
