# round 6: the 16x16x32 rows kernel with G runs at 16-row granularity (product build) against the first version (both row tiles always, ring of
# 12 fragments: -DR16_RT_SKIP=0 -DR16_GK2=3) and against the 32x32x16 kernel: same-box bench lines; parity tests under the product build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j11; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
python -c "from diffdock_pocket_amd import build; build.build(defs=['R16_RT_SKIP=0','R16_GK2=3'], tag='r16noskip', verbose=False)" >> $O/build.log 2>&1; echo "variant rc=$?"
V=$R/diffdock_pocket_amd/libddp_hip_r16noskip.so
line() { python - "$1" "$2" <<PY
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step; rows launch", round(r["avg_launch_ms"],3), "ms; frac", round(r["frac"],4), {k:round(v["avg_launch_ms"],3) for k,v in r["by_layer"].items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for i in 1 2; do
  DDP_ROWS_MFMA16=0 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/b.json 2>> $O/bench.err; line $O/b.json "32x32x16            "
  DDP_ROWS_MFMA16=1 DDP_HIP_LIB=$V timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/b.json 2>> $O/bench.err; line $O/b.json "16x16x32, no skip   "
  DDP_ROWS_MFMA16=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/b.json 2>> $O/bench.err; line $O/b.json "16x16x32, row skip  "
done
DDP_ROWS_MFMA16=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --flex > $O/b.json 2>> $O/bench.err; line $O/b.json "16x16x32 flex       "
DDP_ROWS_MFMA16=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --samples 5 > $O/b.json 2>> $O/bench.err; line $O/b.json "16x16x32 5 samples  "
DDP_ROWS_MFMA16=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --cfg small32 > $O/b.json 2>> $O/bench.err; line $O/b.json "16x16x32 small32    "
DDP_ROWS_MFMA16=0 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --cfg small32 > $O/b.json 2>> $O/bench.err; line $O/b.json "32x32x16 small32    "
DDP_ROWS_MFMA16=1 timeout 1500 python -m pytest tests -m gpu -q -k "single_conv or forward_matches_oracle or every_conv_output or operand_planes or range_flag or capacities or sampler_end_to_end or eliminations or sharing or pruning or pipelined or forked_front or graph_replay" > $O/pytest16.log 2>&1; tail -8 $O/pytest16.log
