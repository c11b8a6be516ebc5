R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j9; mkdir -p $O; cd $R
ulimit -c 0
timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids > $O/stamps_l3.txt; cat $O/stamps_l3.txt
sed -i 's/libddp_hip_rowstamps.so/libddp_hip_rowstamps_g0.so/' tools/stamp_rows.py
timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids > $O/stamps_l3_g0.txt; cat $O/stamps_l3_g0.txt
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for i in 1 2; do
timeout 300 $B > $O/bench.json 2>$O/bench.err; echo "rows 12/6: $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json | head -1)"
DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_g84.so timeout 300 $B > $O/bench84.json 2>$O/bench84.err; echo "rows 8/4: $(grep -o '"ms_per_step": [0-9.]*' $O/bench84.json | head -1)"
DDP_CONV_ROWS=0 timeout 300 $B > $O/bench0.json 2>$O/bench0.err; echo "conv32: $(grep -o '"ms_per_step": [0-9.]*' $O/bench0.json | head -1)"
done
