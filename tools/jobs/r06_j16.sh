# round 6: store patterns of a 3-byte G row (tools/micro/store_g3.hip with the 24-byte-unit modes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j16; mkdir -p $O; cd $R; ulimit -c 0
hipcc --offload-arch=gfx950 -O3 -o $O/store_g3 tools/micro/store_g3.hip > $O/build.log 2>&1; echo "build rc=$?"
timeout 120 $O/store_g3 > $O/store_g3.txt 2>&1; timeout 120 $O/store_g3 >> $O/store_g3.txt 2>&1; cat $O/store_g3.txt; rm -f $O/store_g3
