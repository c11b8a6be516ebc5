R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j24; mkdir -p $O; cd $R
ulimit -c 0
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then D=$R/_ab/old; else D=$R; export DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_noil.so; fi
  [ $v = old ] && unset DDP_HIP_LIB
  cd $D
  for i in 1 2; do timeout 300 python3 $D/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass > $O/b_$v.json 2>$O/b_$v.err; echo "variant [$v]: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$v.json | head -1)"; done
done
done
