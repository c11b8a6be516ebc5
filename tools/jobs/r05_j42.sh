R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j42; mkdir -p $O; cd $R
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "pipelined_layer_order" 2>&1 | tail -2
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for rep in 1 2; do
  timeout 300 $B > $O/b.json 2>$O/b.err; echo "cfg2 default: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
  timeout 300 $B --cfg small32 > $O/b.json 2>$O/b.err; echo "small32 default: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
  timeout 300 $B --cfg small32 --no-split-rows > $O/b.json 2>$O/b.err; echo "small32 one launch: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
  timeout 300 $B --samples 20 > $O/b.json 2>$O/b.err; echo "cfg2 20 samples default: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
  timeout 300 $B --samples 20 --no-split-rows > $O/b.json 2>$O/b.err; echo "cfg2 20 samples one launch: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
done
