R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v2; mkdir -p $O; cd $R
./tools/micro/store_patterns > $O/store_patterns.txt 2>&1; echo "store rc=$?"
./tools/micro/f16x2_mfma > $O/f16x2_mfma.txt 2>&1; echo "f16 rc=$?"
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads"
for v in abl10 abl10_2wg; do
  DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_$v.so timeout 600 $B > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?"
done
for i in 1 2; do
timeout 600 $B --no-roofline-pass > $O/bench_overlap_$i.json 2> $O/bench_overlap_$i.err; echo "overlap rc=$?"
timeout 600 $B --no-roofline-pass --no-overlap-direct > $O/bench_serial_$i.json 2> $O/bench_serial_$i.err; echo "serial rc=$?"
done
timeout 900 python -m pytest tests -m gpu -x -q -k "pipelined" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sel.log
cat $O/store_patterns.txt $O/f16x2_mfma.txt
grep -h -o '"ms_per_step": [0-9.]*' $O/bench_overlap_*.json $O/bench_serial_*.json
