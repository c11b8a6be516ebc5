R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v15; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "stage_a or fp16 or node_enc or single_conv" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sel.log
timeout 300 python tools/bench_stage_a.py 2>&1 | grep -v amdgpu.ids > $O/stage_a.txt; cat $O/stage_a.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads"
for i in 1 2; do timeout 600 $B > $O/bench_$i.json 2> $O/bench.err; echo "bench $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.json | head -2 | tr '\n' ' ')"; done
