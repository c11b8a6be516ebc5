# Everything the round-end evidence comes from, in one gpurun call:  gpurun --timeout 3600 -- "bash tools/gpu_round.sh <name>"
# GPU tests, bench lines (configs[1] with sub-records, configs[2], a 5-sample shard, a 2-rank gloo run), rocprofv3 kernel stats,
# the three PMC passes + tools/pmc_collect.py, device-idle analysis.  Outputs under gpurun_out/<name>/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-round}
mkdir -p $O $O/pmc_fetch $O/pmc_write $O/pmc_mfma
cd $R
if [ -z "$SKIP_PYTEST" ]; then timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log; fi
timeout 400 python bench.py --steps 20 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 200 python bench.py --steps 20 --warmup 2 --flex --no-cpu-baseline > $O/bench_flex.json 2>> $O/bench.err
timeout 200 python bench.py --samples 5 --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_5samples.json 2>> $O/bench.err
timeout 200 env DDP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 1 --no-cpu-baseline --scaling strong 2>> $O/bench.err | grep -v "^\[Gloo\]" > $O/bench_2rank_gloo_strong.json
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-hbm-pass --no-other-workloads"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B > $O/prof.log 2>&1; echo "prof rc=$?"
P="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-pass --no-other-workloads"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $P --launch-log $O/pmc_fetch/launches.json > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $P --launch-log $O/pmc_write/launches.json > $O/pmc_write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $P --launch-log $O/pmc_mfma/launches.json > $O/pmc_mfma.log 2>&1; echo "mfma rc=$?"
# diagnostic passes (wave states, L1 / L2 behaviour of the conv kernels); a counter this build of rocprofv3 does not know fails its pass only
mkdir -p $O/pmc_sq $O/pmc_sq2 $O/pmc_tcp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- $P > $O/pmc_sq.log 2>&1; echo "sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq2 -- $P > $O/pmc_sq2.log 2>&1; echo "sq2 rc=$?"
timeout 600 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_avr --kernel-trace --output-format csv -d $O/pmc_tcp -- $P > $O/pmc_tcp.log 2>&1; echo "tcp rc=$?"
cd $R
for x in sq sq2 tcp; do python3 tools/pmc_raw.py $O/pmc_$x > $O/pmc_$x.json 2>> $O/pmc_raw.err; done
python3 tools/pmc_collect.py $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/r02_pmc.json > $O/pmc_collect.log 2>&1; echo "collect rc=$?"; tail -3 $O/pmc_collect.log
python3 tools/gaps.py $O/prof > $O/gaps.log 2>&1; cat $O/gaps.log | head -12
ls $O $O/prof/* | head -30
# keep the merge small: drop the big traces, keep stats
# keep the merge small (gpurun copies back at most 64 MiB): the traces and raw counter tables are summarised above
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*_agent_info.csv" -delete
du -sh $O
