R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j36; mkdir -p $O; cd $R
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "single_conv_layer or test_every_conv_output or test_forward_matches or small32 or recovered_in_the_same_call or operand_planes" 2>&1 | tail -4
for rep in 1 2; do
for v in new old; do
  D=$R; [ $v = old ] && D=$R/_ab/old
  cd $D
  timeout 300 python3 $D/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass > $O/b_$v.json 2>$O/b_$v.err; echo "variant [$v]: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$v.json | head -1)"
  timeout 300 python3 $D/bench.py --steps 10 --warmup 3 --flex --no-cpu-baseline --no-other-workloads --no-roofline-pass > $O/bf_$v.json 2>$O/bf_$v.err; echo "variant [$v] flex: $(grep -o '"ms_per_step": [0-9.]*' $O/bf_$v.json | head -1)"
done
done
