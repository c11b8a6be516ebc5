R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j26; mkdir -p $O; cd $R
ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "single_conv_layer or test_every_conv_output or test_forward_matches or small32" 2>&1 | tail -2
for rep in 1 2; do
for v in new g8 old; do
  unset DDP_HIP_LIB; D=$R
  [ $v = old ] && D=$R/_ab/old
  [ $v = g8 ] && export DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_g8.so
  cd $D
  timeout 300 python3 $D/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass > $O/b_$v.json 2>$O/b_$v.err; echo "variant [$v]: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$v.json | head -1)"
done
done
cd $R; unset DDP_HIP_LIB
timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids > $O/stamps_l3.txt; cat $O/stamps_l3.txt
