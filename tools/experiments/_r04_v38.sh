R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v38; mkdir -p $O; cd $R
T0=$(date +%s); timeout 200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json | tr '\n' ' ')"
