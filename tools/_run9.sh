F='Wcomment\|^ *[0-9]* |\|^ *|\|warning generated\|In file included\|amdgpu.ids'
for a in 4 3; do
echo "=== DDP_STAMP_ABLATE=$a (4: G pass with 1/4 of its MFMAs and LDS reads; 3: G pass without its global loads)"
DDP_STAMP_ABLATE=$a timeout 400 python tools/stamp_conv.py 2>&1 | grep -v "$F" | sed -n 1,16p
done
