# round 6, end: the whole GPU suite on the final tree (after the task-count cap of the direct convs' split), smoke
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j25; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
