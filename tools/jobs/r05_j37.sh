R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j37; mkdir -p $O; cd $R/tools/micro
timeout 120 ./hbm_write | tee $O/hbm_write.txt
