R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v34; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "forked_front or flexible_layer0 or pruning or clean_pair or graph_replay" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_sel.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
for i in 1 2 3; do for v in fork nofork; do
  F=""; if [ $v = nofork ]; then F="--no-fork-lists-flex"; fi
  timeout 300 $B --samples 4 --cfg cfg1 --flex $F > $O/c1_${v}_$i.json 2> $O/err.txt; echo "cfg1 $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c1_${v}_$i.json | head -1)"
  timeout 300 $B --samples 40 --flex $F > $O/f5_${v}_$i.json 2> $O/err.txt; echo "40 samples flex $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/f5_${v}_$i.json | head -1)"
done; done
