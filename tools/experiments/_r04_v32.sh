R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v32; mkdir -p $O; cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads --cfg small32"
for i in 1 2 3; do for v in base wpe4 wpe5; do
  L=$R/diffdock_pocket_amd/libddp_hip_$v.so; if [ $v = base ]; then L=$R/diffdock_pocket_amd/libddp_hip.so; fi
  DDP_HIP_LIB=$L timeout 300 $B --samples 40 > $O/s40_${v}_$i.json 2> $O/err.txt; echo "small32 x 40 $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/s40_${v}_$i.json | head -1)"
  DDP_HIP_LIB=$L timeout 300 $B --samples 5 > $O/s5_${v}_$i.json 2> $O/err.txt; echo "small32 x 5 $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/s5_${v}_$i.json | head -1)"
done; done
