R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j32; mkdir -p $O; cd $R
ulimit -c 0
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
timeout 300 python tools/stamp_rows.py 2>&1 | grep -v amdgpu.ids > $O/stamps_l3.txt; tail -2 $O/stamps_l3.txt
