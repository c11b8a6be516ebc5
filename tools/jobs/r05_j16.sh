R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j16; mkdir -p $O; cd $R
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "single_conv_layer and 60-10 or test_every_conv_output" 2>&1 | tail -2
for v in rowstamps rs_nobar rs_noepi; do
sed -i "s/libddp_hip_[a-z_]*.so/libddp_hip_$v.so/" tools/stamp_rows.py
echo "== $v"; timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids | grep -E "layer|fc1|seg0|seg1|seg2|workgroup"
done
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
timeout 300 $B > $O/bench.json 2>$O/bench.err; echo "rows: $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json | head -1)"
