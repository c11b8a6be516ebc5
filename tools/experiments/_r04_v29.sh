R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v29; mkdir -p $O; cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads --samples 40 --cfg cfg1"
for i in 1 2 3 4 5 6; do
  for v in default serial snapk; do
    F=""; if [ $v = serial ]; then F="--no-overlap-direct"; fi; if [ $v = snapk ]; then F="--snapshot-kernel"; fi
    timeout 300 $B $F > $O/${v}_$i.json 2> $O/err.txt; echo "cfg1 x 40 rigid $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/${v}_$i.json | head -1)"
  done
done
