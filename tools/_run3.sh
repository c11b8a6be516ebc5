ulimit -c 0
O=gpurun_out/r03_f4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_lists.py tests/test_gpu_parity.py -m gpu -q -x -k "flex_mark or clean_pair or flexible_layer0 or graph_replay or layer0_sharing or sampler_end or forward_matches or capacities" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
for a in "--flex --samples 4 --cfg cfg1" "--flex" "--flex --samples 5"; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads $a 2>>$O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a |', round(d['ms_per_step'],3), '|', round(d['value'],2))"; done
