# Everything the round-end evidence comes from, in one gpurun call:  gpurun --timeout 3600 -- "bash tools/gpu_round.sh <name>"
# GPU tests, bench lines (configs[1] with sub-records + cpu_baseline, configs[2], the 5-sample shard, a 2-rank gloo run), rocprofv3
# kernel stats (rigid, flexible, 5 samples, cfg1), the three PMC passes + tools/pmc_collect.py, device-idle analysis.
# Outputs under gpurun_out/<name>/.   SKIP_PYTEST=1 / SKIP_CPU=1 / SKIP_PMC=1 shorten it, CPU_FULL=1 adds the unbounded CPU baseline.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-round}
mkdir -p $O $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_l1l2 $O/pmc_lat $O/pmc_sq
cd $R
ulimit -c 0
if [ -z "$SKIP_PYTEST" ]; then timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log; fi
CPU=""; if [ -n "$SKIP_CPU" ]; then CPU="--no-cpu-baseline"; fi
timeout 1500 python bench.py --steps 20 --warmup 3 $CPU > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
if [ -n "$CPU_FULL" ]; then timeout 1500 python bench.py --steps 20 --warmup 3 --no-other-workloads --no-roofline-pass --cpu-full > $O/cpu_baseline_full.json 2>> $O/bench.err; echo "cpu full rc=$?"; fi
timeout 300 python bench.py --steps 20 --warmup 3 --flex --no-cpu-baseline > $O/bench_flex.json 2>> $O/bench.err
timeout 300 python bench.py --samples 5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_5samples.json 2>> $O/bench.err
timeout 300 env DDP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline 2>> $O/bench.err | grep -v "^\[Gloo\]" > $O/bench_2rank_gloo_strong.json
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B > $O/prof.log 2>&1; echo "prof rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_flex -- $B --flex > $O/prof_flex.log 2>&1; echo "prof flex rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_5 -- $B --samples 5 > $O/prof_5.log 2>&1; echo "prof 5 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg1 -- $B --samples 4 --cfg cfg1 --flex > $O/prof_cfg1.log 2>&1; echo "prof cfg1 rc=$?"
if [ -z "$SKIP_PMC" ]; then
P="python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-hbm-pass --no-other-workloads"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $P --launch-log $O/pmc_fetch/launches.json > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $P --launch-log $O/pmc_write/launches.json > $O/pmc_write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $P --launch-log $O/pmc_mfma/launches.json > $O/pmc_mfma.log 2>&1; echo "mfma rc=$?"
cd $R
cd /tmp
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_l1l2 -- $P --no-roofline-pass > $O/pmc_l1l2.log 2>&1; echo "l1l2 rc=$?"
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_lat -- $P --no-roofline-pass > $O/pmc_lat.log 2>&1; echo "lat rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -- $P --no-roofline-pass > $O/pmc_sq.log 2>&1; echo "sq rc=$?"
cd $R
python3 tools/pmc_raw.py $O/pmc_sq > $O/r06_pmc_wave_states.json 2>/dev/null
python3 tools/pmc_collect.py $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/r06_pmc.json $O/pmc_l1l2 $O/pmc_lat > $O/pmc_collect.log 2>&1; echo "collect rc=$?"; tail -3 $O/pmc_collect.log
fi
# the default bench line once more with THIS run's counter file (DDP_PMC_FILE; the tree's profiles/r06_pmc.json is only replaced by hand, from gpurun_out)
if [ -s $O/r06_pmc.json ]; then DDP_PMC_FILE=$O/r06_pmc.json timeout 900 python bench.py --steps 20 --warmup 3 --cpu-budget-s 30 > $O/bench_pmc.json 2>> $O/bench.err; echo "bench (fresh pmc) rc=$?"; fi
cd $R
for d in prof prof_flex prof_5 prof_cfg1; do echo "== $d"; python3 tools/gaps.py $O/$d 2>&1 | head -6; python3 tools/step_sequence.py $O/$d > $O/$d.sequence.txt 2>&1; tail -1 $O/$d.sequence.txt; done > $O/gaps.log 2>&1
cat $O/gaps.log
# keep the merge small (gpurun copies back at most 64 MiB): the traces and raw counter tables are summarised above
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*_agent_info.csv" -delete
du -sh $O
