R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j41; mkdir -p $O; cd $R
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "pipelined_layer_order or forked_front or capacity or test_forward_matches or recovered_in_the_same_call" 2>&1 | tail -3
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for rep in 1 2; do
  timeout 300 $B > $O/b.json 2>$O/b.err; echo "default (split): $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
  timeout 300 $B --no-split-rows > $O/b2.json 2>$O/b2.err; echo "one launch: $(grep -o '"ms_per_step": [0-9.]*' $O/b2.json | head -1)"
done
timeout 300 $B --samples 5 > $O/b.json 2>$O/b.err; echo "5 samples: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
