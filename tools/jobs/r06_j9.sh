# round 6: the rows kernel's stream-tile loop on v_mfma_f32_16x16x32_f16 against the shipped 32x32x16 form (tools/micro/stream_16.hip)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j9; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O3 -o tools/micro/stream_16 tools/micro/stream_16.hip > $O/build.log 2>&1; echo "build rc=$?"
./tools/micro/stream_16 > $O/stream_16.txt 2>&1; cat $O/stream_16.txt
./tools/micro/stream_16 >> $O/stream_16.txt 2>&1; tail -8 $O/stream_16.txt
