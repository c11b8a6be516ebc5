R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j31; mkdir -p $O; cd $R
ulimit -c 0
timeout 600 python tools/two_graphs.py 2 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/two.txt
timeout 600 python tools/two_graphs.py 4 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/four.txt
