R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j39; mkdir -p $O; cd $R
ulimit -c 0
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for rep in 1 2; do
  timeout 300 $B > $O/b.json 2>$O/b.err; echo "pipelined (default): $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
  timeout 300 $B --no-overlap-direct > $O/b.json 2>$O/b.err; echo "serial (--no-overlap-direct): $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
  timeout 300 $B --flex > $O/b.json 2>$O/b.err; echo "flex pipelined: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
  timeout 300 $B --flex --no-overlap-direct > $O/b.json 2>$O/b.err; echo "flex serial: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
done
