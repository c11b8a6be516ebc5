# round 6: ddp_conv_rows on v_mfma_f32_16x16x32_f16 (model.rows_mfma16, csrc/ddp_conv_rows16.hip): parity tests under it, same-box bench pairs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j10; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
DDP_ROWS_MFMA16=1 timeout 1500 python -m pytest tests -m gpu -q -k "single_conv or forward_matches_oracle or every_conv_output or operand_planes or range_flag or capacities or sampler_end_to_end" > $O/pytest16.log 2>&1; tail -25 $O/pytest16.log
for f in 0 1 0 1; do
  DDP_ROWS_MFMA16=$f timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/bench_$f.json 2>> $O/bench.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("rows_mfma16=$f", round(d["value"],2), "poses/s", round(d["ms_per_step"],3), "ms/step; rows launch", round(r["avg_launch_ms"],3), "ms; frac", round(r["frac"],4), {k:round(v["avg_launch_ms"],3) for k,v in r["by_layer"].items()})
except Exception as e:
    print("rows_mfma16=$f failed", e)
PY
done
tail -5 $O/bench.err
