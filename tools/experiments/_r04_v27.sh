R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v27; mkdir -p $O; cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads"
for i in 1 2 3; do for v in forked large; do
  F=""; if [ $v = large ]; then F="--concurrent-max-atoms 0"; fi
  timeout 600 $B --samples 5 $F > $O/b5_${v}_$i.json 2> $O/b5_$v.err; echo "5 samples $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/b5_${v}_$i.json | head -1)"
  timeout 600 $B --samples 4 --cfg cfg1 --flex $F > $O/c1_${v}_$i.json 2> $O/c1_$v.err; echo "cfg1 $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c1_${v}_$i.json | head -1)"
  timeout 600 $B --samples 5 --flex $F > $O/f5_${v}_$i.json 2> $O/f5_$v.err; echo "5 samples flex $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/f5_${v}_$i.json | head -1)"
done; done
