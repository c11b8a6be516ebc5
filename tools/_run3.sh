ulimit -c 0
O=gpurun_out/r03_f2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "flexible_layer0 or layer1_clean or layer0_sharing or capacities or forward_matches or bench_batch or sampler_end or device_driven or pruning or every_conv" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
