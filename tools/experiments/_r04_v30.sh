R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v30; mkdir -p $O; cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
for i in 1 2 3; do for v in upload skip; do
  E=""; if [ $v = skip ]; then export DDP_TIMING_SKIP_UPLOAD=1; else unset DDP_TIMING_SKIP_UPLOAD; fi
  timeout 300 $B --samples 5 > $O/b5_${v}_$i.json 2> $O/err.txt; echo "5 samples $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/b5_${v}_$i.json | head -1)"
  timeout 300 $B --samples 4 --cfg cfg1 --flex > $O/c1_${v}_$i.json 2> $O/err.txt; echo "cfg1 $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c1_${v}_$i.json | head -1)"
done; done
