R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j33; mkdir -p $O; cd $R/tools/micro
timeout 300 ./stream_wide | tee $O/stream_wide.txt
