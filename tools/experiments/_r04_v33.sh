R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v33; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "forked_front" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_sel.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
for i in 1 2 3; do for v in fork nofork; do
  F=""; if [ $v = nofork ]; then F="--no-fork-means"; fi
  timeout 300 $B --samples 5 $F > $O/b5_${v}_$i.json 2> $O/err.txt; echo "5 samples $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/b5_${v}_$i.json | head -1)"
  timeout 300 $B --samples 4 --cfg cfg1 --flex $F > $O/c1_${v}_$i.json 2> $O/err.txt; echo "cfg1 $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c1_${v}_$i.json | head -1)"
  timeout 300 $B --samples 5 --flex $F > $O/f5_${v}_$i.json 2> $O/err.txt; echo "5 samples flex $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/f5_${v}_$i.json | head -1)"
done; done
