R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v5; mkdir -p $O; cd $R
timeout 600 python tools/stamp_conv.py > $O/stamps_h2.txt 2>&1; echo "stamps rc=$?"
for pad in 56 100; do DDP_STAMP_LDS_PAD_KB=$pad timeout 600 python tools/stamp_conv.py > $O/stamps_h2_pad$pad.txt 2>&1; done
head -20 $O/stamps_h2.txt; head -20 $O/stamps_h2_pad100.txt
