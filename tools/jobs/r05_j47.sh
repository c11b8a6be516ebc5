R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j47; mkdir -p $O; cd $R
ulimit -c 0
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for rep in 1 2; do
  timeout 300 $B > $O/b.json 2>$O/b.err; echo "default: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
  timeout 300 $B --stage-a-first > $O/b2.json 2>$O/b2.err; echo "stage A of the atom rows queued first: $(grep -o '"ms_per_step": [0-9.]*' $O/b2.json | head -1)"; tail -1 $O/b2.err | grep -v amdgpu
  timeout 300 $B --flex > $O/b.json 2>$O/b.err; echo "flex default: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json | head -1)"
  timeout 300 $B --flex --stage-a-first > $O/b2.json 2>$O/b2.err; echo "flex stage A first: $(grep -o '"ms_per_step": [0-9.]*' $O/b2.json | head -1)"
done
