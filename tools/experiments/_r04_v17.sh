R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v17; mkdir -p $O; cd $R
for m in plain reset close nograph; do timeout 600 python tools/graph_mem.py $m 2>&1 | grep -v amdgpu.ids; done > $O/graph_mem.txt 2>&1
cat $O/graph_mem.txt
