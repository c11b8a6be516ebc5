R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v1; mkdir -p $O; cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads"
for v in base abl10 abl10_2wg base_2wg; do
  L=$R/diffdock_pocket_amd/libddp_hip_$v.so; if [ $v = base ]; then L=$R/diffdock_pocket_amd/libddp_hip.so; fi
  DDP_HIP_LIB=$L timeout 600 $B > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?"
done
timeout 600 python tools/traj_divergence.py > $O/traj.txt 2>&1; echo "traj rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q -k "hetero or small32 or bench_batch_samples or every_conv or capacities" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sel.log
