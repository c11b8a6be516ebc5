R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v0; mkdir -p $O; cd $R
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 600 $O/bench.json
