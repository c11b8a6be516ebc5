R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v7; mkdir -p $O; cd $R
for v in base sa4 sa5 sa6; do
  L=$R/diffdock_pocket_amd/libddp_hip_$v.so; if [ $v = base ]; then L=$R/diffdock_pocket_amd/libddp_hip.so; fi
  echo "== $v"; DDP_HIP_LIB=$L timeout 300 python tools/bench_stage_a.py 2>&1 | grep -v amdgpu.ids
done > $O/stage_a_abl.txt 2>&1
cat $O/stage_a_abl.txt
