# round 6, end: the tests that build stand-alone conv layers, after they took the shipped kernel forms as defaults; then the whole suite once more
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j27; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(round(d["value"],2), round(d["ms_per_step"],3), r["kernel"], round(r["frac"],4), r.get("padding_frac_pmc"), r.get("traffic"), (r.get("pmc_source") or {}).get("stale"), d["scaling"])
PY
