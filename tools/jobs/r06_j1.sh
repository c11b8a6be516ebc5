# round 6, first GPU call: fp8 conversion semantics, the four overlap numbers of VERDICT item 1, a parity subset on the ABI-16 tree, the bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j1; mkdir -p $O; cd $R; ulimit -c 0
./tools/micro/fp8_cvt > $O/fp8_cvt.txt 2>&1; echo "fp8 rc=$?"
timeout 900 python tools/overlap_ab.py > $O/overlap_ab.txt 2> $O/overlap_ab.err; echo "overlap rc=$?"; cat $O/overlap_ab.txt; tail -5 $O/overlap_ab.err
timeout 1500 python -m pytest tests -m gpu -x -q -k "sampler_end_to_end or forward_matches_oracle or every_conv_output or stage_a or single_conv or recovered" > $O/pytest_subset.log 2>&1; tail -5 $O/pytest_subset.log
timeout 900 python bench.py --steps 20 --warmup 3 --cpu-budget-s 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"].get("also_measured"))
PY
