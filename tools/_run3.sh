ulimit -c 0
O=gpurun_out/r03_h9; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "graph_replay or forward_matches or capacities or deterministic or sampler_end or device_driven or every_conv or trajectory" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for a in "--samples 4 --cfg cfg1 --flex" "--samples 5" "" "--flex"; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads $a 2>>$O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['ms_per_step'], d['value'])"; done
