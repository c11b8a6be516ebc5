R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v23; mkdir -p $O; cd $R
for v in base sapin; do
  L=$R/diffdock_pocket_amd/libddp_hip_$v.so; if [ $v = base ]; then L=$R/diffdock_pocket_amd/libddp_hip.so; fi
  echo "== $v"; DDP_HIP_LIB=$L timeout 300 python tools/bench_stage_a.py 2>&1 | grep -v amdgpu.ids
done > $O/stage_a.txt 2>&1
cat $O/stage_a.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for i in 1 2; do for v in base sapin; do
  L=$R/diffdock_pocket_amd/libddp_hip_$v.so; if [ $v = base ]; then L=$R/diffdock_pocket_amd/libddp_hip.so; fi
  DDP_HIP_LIB=$L timeout 600 $B > $O/bench_${v}_$i.json 2> $O/bench_$v.err; echo "$v $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$i.json | head -1)"
done; done
