R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v16; mkdir -p $O; cd $R
rm -f $O/mem.log
DDP_TEST_MEM_LOG=$O/mem.log timeout 1500 python -m pytest tests -m gpu -q -k "pipelined or bench_batch or graph_replay or fp16_split or flexible_layer0" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sel.log
cat $O/mem.log
bash tools/_r04_v15.sh
