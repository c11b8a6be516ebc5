# round 6, end-of-round evidence on the FINAL tree (19-bit plane form of G, direct convs through the row-stationary kernel): everything
# tools/gpu_round.sh collects, then smoke() and the plain default bench line as the driver runs it
R=$GRAFT_REPO_ROOT; cd $R
python -m diffdock_pocket_amd.build > /dev/null 2>&1; echo "build rc=$?"
bash tools/gpu_round.sh r06_final5
O=$R/gpurun_out/r06_final5
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
cp $O/r06_pmc.json $R/profiles/r06_pmc.json
timeout 900 python bench.py > $O/bench_default_flags.json 2> $O/bench_default.err; echo "default bench rc=$?"
