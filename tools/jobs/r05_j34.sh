R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j34; mkdir -p $O; cd $R/tools/micro
timeout 60 ./mfma_k8 | tee $O/mfma_k8.txt
