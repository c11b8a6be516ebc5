R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j18; mkdir -p $O; cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "single_conv_layer or test_every_conv_output or small32 or error_floor or stage_a" 2>&1 | tail -4
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
timeout 300 $B --cfg small32 > $O/b_small.json 2>$O/b_small.err; echo "small32 rows: $(grep -o '"ms_per_step": [0-9.]*' $O/b_small.json | head -1)"
DDP_CONV_ROWS=0 timeout 300 $B --cfg small32 > $O/b_small0.json 2>$O/b_small0.err; echo "small32 conv32: $(grep -o '"ms_per_step": [0-9.]*' $O/b_small0.json | head -1)"
timeout 300 $B --samples 5 > $O/b5.json 2>$O/b5.err; echo "5 samples rows: $(grep -o '"ms_per_step": [0-9.]*' $O/b5.json | head -1)"
DDP_CONV_ROWS=0 timeout 300 $B --samples 5 > $O/b50.json 2>$O/b50.err; echo "5 samples conv32: $(grep -o '"ms_per_step": [0-9.]*' $O/b50.json | head -1)"
timeout 300 $B --flex > $O/bf.json 2>$O/bf.err; echo "flex rows: $(grep -o '"ms_per_step": [0-9.]*' $O/bf.json | head -1)"
timeout 300 $B > $O/b.json 2>$O/b.err; echo "rigid rows: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
