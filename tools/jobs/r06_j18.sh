# round 6, late: g_planes3 (19-bit plane form 1) as the default - the whole GPU suite, smoke, parity of both forms for the record, the default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j18; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python tools/g3_parity.py > $O/g19.txt 2>&1; tail -8 $O/g19.txt | cut -c1-330
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(round(d["value"],2), round(d["ms_per_step"],3), r["kernel"], round(r["frac"],4), r.get("padding_frac_pmc"), (r.get("pmc_source") or {}).get("stale"), d["scaling"], d["config"]["also_measured"])
PY
