R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v37; mkdir -p $O; cd $R
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
for i in 1 2 3 4 5; do timeout 120 $B > $O/b_$i.json 2> $O/err.txt; echo "default rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/b_$i.json | head -1)"; done
