R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j27; mkdir -p $O; cd $R
ulimit -c 0
for v in rowstamps sta1 sta2; do
  export STAMP_LIB=$R/diffdock_pocket_amd/libddp_hip_$v.so
  timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids > $O/stamps_$v.txt; echo "== $v"; grep -E "layer|features|seg1 stream|workgroup" $O/stamps_$v.txt
done
