# round 5, job 14: counters of ddp_conv_rows_kernel (wave states, L1/L2, latencies) + the three bench.py PMC passes -> r05_pmc.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j14; mkdir -p $O $O/pmc_fetch $O/pmc_write $O/pmc_mfma; cd /tmp; export TMPDIR=/tmp
ulimit -c 0
P="python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-hbm-pass --no-other-workloads"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -- $P --no-roofline-pass > $O/pmc_sq.log 2>&1; echo "sq rc=$?"
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_l1l2 -- $P --no-roofline-pass > $O/pmc_l1l2.log 2>&1; echo "l1l2 rc=$?"
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_BUSY_avr GRBM_GUI_ACTIVE TCC_EA0_RD_LATENCY_sum --kernel-trace --output-format csv -d $O/pmc_lat -- $P --no-roofline-pass > $O/pmc_lat.log 2>&1; echo "lat rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $P --launch-log $O/pmc_fetch/launches.json > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $P --launch-log $O/pmc_write/launches.json > $O/pmc_write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $P --launch-log $O/pmc_mfma/launches.json > $O/pmc_mfma.log 2>&1; echo "mfma rc=$?"
cd $R
for d in pmc_sq pmc_l1l2 pmc_lat; do python3 tools/pmc_raw.py $O/$d > $O/$d.json 2> $O/$d.err; done
python3 tools/pmc_collect.py $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/r05_pmc.json > $O/pmc_collect.log 2>&1; echo "collect rc=$?"; tail -3 $O/pmc_collect.log
python3 -c "
import json
for f in ('pmc_sq','pmc_l1l2','pmc_lat'):
    d=json.load(open('$O/'+f+'.json')); print(f, json.dumps(d.get('ddp_conv_rows_kernel')))
"
find $O -name "*counter_collection.csv" -size +2M -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*_agent_info.csv" -delete
