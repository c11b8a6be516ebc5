R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v36; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pipelined_layer_order" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sel.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
for i in 1 2 3; do for v in pipeline4 pipeline5; do
  timeout 300 $B --layer-order $v > $O/b_${v}_$i.json 2> $O/err.txt; echo "$v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/b_${v}_$i.json | head -1)"
  timeout 300 $B --flex --layer-order $v > $O/f_${v}_$i.json 2> $O/err.txt; echo "$v flex rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/f_${v}_$i.json | head -1)"
done; done
