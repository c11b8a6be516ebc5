# timing-only ablation builds under the in-kernel stamps: bash tools/abl_stamps.sh "<ablation ids>" "<LDS pad KB list>"
mkdir -p gpurun_out/abl
for a in ${1:-1 3 7}; do
  for pad in ${2:-0}; do
    DDP_STAMP_ABLATE=$a DDP_STAMP_LDS_PAD_KB=$pad timeout 250 python tools/stamp_conv.py > gpurun_out/abl/stamp_abl${a}_pad${pad}.txt 2>&1
    echo "== ablate $a pad $pad"; grep "HIP-event\|stage edge\|fc1 \|features\|G pass (wave 0)\|role tiles\|park\|mean total" gpurun_out/abl/stamp_abl${a}_pad${pad}.txt
  done
done
