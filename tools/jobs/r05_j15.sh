R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j15; mkdir -p $O; cd $R
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "single_conv_layer and 60-10 or test_every_conv_output" 2>&1 | tail -2
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for i in 1 2 3; do
timeout 300 $B > $O/bench.json 2>$O/bench.err; echo "rows no-slp: $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json | head -1)"
DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_slp.so timeout 300 $B > $O/bench1.json 2>$O/bench1.err; echo "rows slp: $(grep -o '"ms_per_step": [0-9.]*' $O/bench1.json | head -1)"
done
timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids > $O/stamps_l3.txt; cat $O/stamps_l3.txt
