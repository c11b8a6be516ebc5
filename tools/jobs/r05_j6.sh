R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j6; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_every_conv_output and cfg2_full_noflex" 2>&1 | grep -E "assert|Error|error|conv|FAILED" | head -30
for a in "3 200 23" "1 300 40" "0 129 5" "3 3000 37"; do echo "== $a"; timeout 120 python tools/dbg_rows.py $a 2>&1 | grep -v amdgpu.ids | head -6; done
