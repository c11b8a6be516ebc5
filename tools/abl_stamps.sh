mkdir -p gpurun_out/r2m
for a in 3 4 5 6; do
  for pad in 60 0; do
    DDP_STAMP_ABLATE=$a DDP_STAMP_LDS_PAD_KB=$pad timeout 250 python tools/stamp_conv.py > gpurun_out/r2m/stamp_abl${a}_pad${pad}.txt 2>&1
    echo "== ablate $a pad $pad"; grep "HIP-event\|G pass (wave 0)\|role tiles\|mean total\|g_stage wave 0" gpurun_out/r2m/stamp_abl${a}_pad${pad}.txt
  done
done
