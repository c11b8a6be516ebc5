# round 6, late: single-conv tests through both kernel forms / plane forms / the direct form; the form-0 kernel reading plane form 1 (model level)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j23; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q -k "single_conv or small_magnitudes" > $O/pytest_single.log 2>&1; echo "single conv: $(tail -1 $O/pytest_single.log)"; grep -n "^FAILED" $O/pytest_single.log | head -20
DDP_ROWS_MFMA16=0 timeout 900 python -m pytest tests -m gpu -q -k "forward_matches_oracle or every_conv_output" > $O/pytest_form0_g3.log 2>&1; echo "rows_mfma16=0 (g_planes3=1): $(tail -1 $O/pytest_form0_g3.log)"
