R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v31; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_inference_csv.py -x -q -m gpu -k "prefetched or graph_replay or sampler or csv or device_driven or captured" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sel.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
for i in 1 2 3; do for v in prefetch upload; do
  F=""; if [ $v = upload ]; then F="--no-prefetch"; fi
  timeout 300 $B --samples 5 $F > $O/b5_${v}_$i.json 2> $O/err.txt; echo "5 samples $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/b5_${v}_$i.json | head -1)"
  timeout 300 $B --samples 4 --cfg cfg1 --flex $F > $O/c1_${v}_$i.json 2> $O/err.txt; echo "cfg1 $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c1_${v}_$i.json | head -1)"
  timeout 300 $B $F > $O/b40_${v}_$i.json 2> $O/err.txt; echo "40 samples $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/b40_${v}_$i.json | head -1)"
done; done
