R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fp16_range" 2>&1 | grep -vE "^\s*$" | tail -60
