# round 6: single-conv cases with vector blocks wider than 16 columns, both forms of the row-stationary kernel; the option tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j13; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
DDP_ROWS_MFMA16=1 timeout 600 python -m pytest tests -m gpu -q -k "single_conv" > $O/pytest16.log 2>&1; tail -6 $O/pytest16.log
DDP_ROWS_MFMA16=0 timeout 600 python -m pytest tests -m gpu -q -k "single_conv" > $O/pytest32.log 2>&1; tail -6 $O/pytest32.log
timeout 900 python tools/g3_parity.py > $O/g3.txt 2>&1; tail -4 $O/g3.txt
