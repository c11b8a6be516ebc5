R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v24; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "forked_front or pipelined_layer_order or graph_replay or determin" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sel.log
DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_satune.so timeout 300 python tools/bench_stage_a_rows.py 2>&1 | grep -v amdgpu.ids > $O/stage_a_rows.txt; cat $O/stage_a_rows.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass"
for i in 1 2; do for v in base nofork; do
  F=""; if [ $v = nofork ]; then F="--no-fork-front"; fi
  timeout 900 $B $F > $O/bench_${v}_$i.json 2> $O/bench_$v.err; echo "$v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$i.json | tr '\n' ' ')"
done; done
