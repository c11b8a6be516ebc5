R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j1; mkdir -p $O; cd $R
ulimit -c 0
for a in "3 200 23" "1 300 40" "3 3000 37"; do echo "== $a"; timeout 120 python tools/dbg_rows.py $a 2>&1 | grep -v amdgpu.ids | head -8; done > $O/dbg.txt 2>&1; cat $O/dbg.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "single_conv_layer and 60-10" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "test_forward_matches_oracle_and_golden or test_every_conv_output" 2>&1 | tail -15
