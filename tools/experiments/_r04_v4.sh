R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v4; mkdir -p $O; cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads"
for v in base r4 r8 nob nog nobg w2r8; do
  L=$R/diffdock_pocket_amd/libddp_hip_$v.so; if [ $v = base ]; then L=$R/diffdock_pocket_amd/libddp_hip.so; fi
  DDP_HIP_LIB=$L timeout 600 $B > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.json | head -2 | tr '\n' ' ')"
done
