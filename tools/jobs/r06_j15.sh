# round 6: precision of a 19-bit G plane (fp16 hi + a byte relative to hi's own ulp) BEFORE building it: stage A of a variant library rounds the
# fp16 lo word to that grid (-DDDP_GH_LO19), everything else as shipped; tools/g3_parity.py --forms 0 under that library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j15; mkdir -p $O; cd $R; ulimit -c 0
python -m diffdock_pocket_amd.build > $O/build.log 2>&1; echo "build rc=$?"
python -c "from diffdock_pocket_amd import build; print(build.build(defs=['DDP_GH_LO19=1'], tag='lo19'))" >> $O/build.log 2>&1; echo "variant rc=$?"
DDP_HIP_LIB=$R/diffdock_pocket_amd/libddp_hip_lo19.so timeout 1100 python tools/g3_parity.py --forms 0 > $O/lo19.txt 2>&1; tail -12 $O/lo19.txt | cut -c1-400
