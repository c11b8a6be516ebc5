ulimit -c 0
O=gpurun_out/r03_f3; mkdir -p $O
for a in "--flex" "--flex --no-flex-sharing" "--flex" "--flex --no-flex-sharing" "--flex --samples 5" "--flex --samples 5 --no-flex-sharing" "--flex --samples 4 --cfg cfg1" "--flex --samples 4 --cfg cfg1 --no-flex-sharing"; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-workloads $a 2>>$O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a |', round(d['ms_per_step'],3), '|', round(d['value'],2))"; done 2>&1 | tee $O/ab.txt
