R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v28; mkdir -p $O; cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
for i in 1 2 3; do
  for n in 15 20; do timeout 300 $B --samples $n > $O/n${n}_$i.json 2> $O/err.txt; echo "cfg2 $n samples rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/n${n}_$i.json | head -1)"; done
  timeout 300 $B --samples 40 --cfg cfg1 --flex > $O/c1_40_$i.json 2> $O/err.txt; echo "cfg1 40 samples flex rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c1_40_$i.json | head -1)"
  timeout 300 $B --samples 40 --cfg cfg1 > $O/c1r_40_$i.json 2> $O/err.txt; echo "cfg1 40 samples rigid rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c1r_40_$i.json | head -1)"
  for o in pipeline pipeline3 pipeline4; do timeout 300 $B --samples 4 --cfg cfg1 --flex --concurrent-max-atoms 0 --layer-order $o > $O/c1_${o}_$i.json 2> $O/err.txt; echo "cfg1 4 samples large-batch order $o rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c1_${o}_$i.json | head -1)"; done
done
