# round 6, late: the parity subset under the non-default combinations of the three switches (form-0 rows kernel reading plane form 1, 4-byte planes, direct convs on ddp_conv_messages)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j22; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
K="forward_matches_oracle or every_conv_output or single_conv or rows_kernel_range_flag or cfg2_job_end_to_end"
DDP_ROWS_MFMA16=0 timeout 900 python -m pytest tests -m gpu -q -k "$K" > $O/pytest_form0_g3.log 2>&1; echo "rows_mfma16=0 (g_planes3=1): $(tail -1 $O/pytest_form0_g3.log)"
DDP_ROWS_MFMA16=0 DDP_G_PLANES3=0 timeout 900 python -m pytest tests -m gpu -q -k "$K" > $O/pytest_form0_g4.log 2>&1; echo "rows_mfma16=0 g_planes3=0: $(tail -1 $O/pytest_form0_g4.log)"
DDP_G_PLANES3=0 DDP_DIRECT_ROWS=0 timeout 900 python -m pytest tests -m gpu -q -k "$K" > $O/pytest_g4_nodirect.log 2>&1; echo "g_planes3=0 direct_rows=0: $(tail -1 $O/pytest_g4_nodirect.log)"
