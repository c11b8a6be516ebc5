R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_final5; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
grep -o '"value": [0-9.]*' $O/bench.json | head -1
