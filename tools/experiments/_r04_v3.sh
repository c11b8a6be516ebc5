R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v3; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fp16_split or single_conv or forward_matches or every_conv" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_sel.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads"
timeout 600 $B > $O/bench_h2.json 2> $O/bench_h2.err; echo "bench rc=$?"
grep -o '"ms_per_step": [0-9.]*' $O/bench_h2.json | head -3
