# round 6: the 3-byte G plane's parity margins (tools/g3_parity.py), the pair's start / end offsets under rocprofv3, the whole GPU suite (default form) with durations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j4; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
timeout 1200 python tools/g3_parity.py > $O/g3_parity.txt 2> $O/g3_parity.err; echo "g3 parity rc=$?"; cat $O/g3_parity.txt; tail -3 $O/g3_parity.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/pair_trace -- python3 $R/tools/overlap_ab.py --pair-only > $O/pair_trace.log 2>&1; echo "pair trace rc=$?"
cd $R
python3 tools/pair_offsets.py $O/pair_trace > $O/pair_offsets.txt 2>&1; cat $O/pair_offsets.txt
find $O/pair_trace -name "*.csv" -size +1M -delete
timeout 2400 python -m pytest tests -m gpu -q --durations=30 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -45 $O/pytest.log
