R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j45; mkdir -p $O; cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "single_conv_layer or test_every_conv_output or test_forward_matches" 2>&1 | tail -2
for v in new noxcd new noxcd; do
  unset DDP_HIP_LIB; [ $v != new ] && export DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_$v.so
  echo "== $v"; timeout 300 python tools/bench_stage_a.py 2>&1 | grep -E "plane" | grep -v amdgpu
done
for rep in 1 2; do
for v in new noxcd; do
  unset DDP_HIP_LIB; [ $v != new ] && export DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_$v.so
  timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass > $O/b_$v.json 2>$O/b_$v.err; echo "variant [$v]: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$v.json | head -1)"
done
done
