# round 6, end-of-round evidence on the final tree: everything tools/gpu_round.sh collects (GPU suite, bench lines with sub-records and cpu_baseline,
# the unbounded CPU baseline of SURVEY 8(d), configs[2], the 5-sample shard, a 2-rank gloo run, rocprofv3 kernel stats x 4, the PMC passes,
# the bench line once more with this run's counters), then smoke() and the plain default bench line as the driver runs it
R=$GRAFT_REPO_ROOT; cd $R
python -m diffdock_pocket_amd.build > /dev/null 2>&1; echo "build rc=$?"
CPU_FULL=1 bash tools/gpu_round.sh r06_final
O=$R/gpurun_out/r06_final
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
cp $O/r06_pmc.json $R/profiles/r06_pmc.json
timeout 900 python bench.py > $O/bench_default_flags.json 2> $O/bench_default.err; echo "default bench rc=$?"
