R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v10; mkdir -p $O; cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --no-roofline-pass"
for i in 1 2; do
for v in chains pipeline pipeline2; do timeout 600 $B --layer-order $v > $O/bench_${v}_$i.json 2> $O/err_$v.txt; echo "$v $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${v}_$i.json | head -1)"; done
timeout 600 $B --no-overlap-direct > $O/bench_serial_$i.json 2> $O/err_serial.txt; echo "serial $(grep -o '"ms_per_step": [0-9.]*' $O/bench_serial_$i.json | head -1)"
done
timeout 900 python -m pytest tests -m gpu -q -x -k "pipelined" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sel.log
