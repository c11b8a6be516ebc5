R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v14; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
