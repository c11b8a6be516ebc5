R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v35; mkdir -p $O; cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads --samples 4 --cfg cfg1 --flex"
export DDP_UPLOAD_TIMING=1
for i in 1 2 3 4 5 6 7 8; do for v in ring fresh; do
  if [ $v = fresh ]; then export DDP_PIN_RING=0; else export DDP_PIN_RING=1; fi
  timeout 300 $B > $O/c1_${v}_$i.json 2> $O/err_${v}_$i.txt; echo "cfg1 $v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c1_${v}_$i.json | head -1) $(grep 'host us' $O/err_${v}_$i.txt | tail -1)"
done; done
