R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j21; mkdir -p $O; cd $R
ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "single_conv_layer or test_every_conv_output or test_forward_matches or recovered_in_the_same_call or small32" 2>&1 | tail -5
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for i in 1 2; do timeout 300 $B > $O/b.json 2>$O/b.err; echo "rigid rows: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"; done
timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids > $O/stamps_l3.txt; cat $O/stamps_l3.txt
