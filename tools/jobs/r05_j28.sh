R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j28; mkdir -p $O; cd $R/tools/micro
timeout 120 ./mfma_chain | tee $O/mfma_chain.txt
