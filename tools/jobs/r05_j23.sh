R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j23; mkdir -p $O; cd $R
ulimit -c 0
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for v in noil stag3 stag8 stag0 noil; do
  export DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_$v.so
  [ ! -f "$DDP_HIP_LIB" ] && continue
  for i in 1 2; do timeout 300 $B > $O/b.json 2>$O/b.err; echo "variant [$v]: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"; done
done
