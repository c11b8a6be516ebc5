R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v18; mkdir -p $O; cd $R
rm -f $O/mem.log
DDP_TEST_MEM_LOG=$O/mem.log timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
awk -F'\t' '{print $2, $3}' $O/mem.log | sort | uniq -c | sort -k2 | tail -12
