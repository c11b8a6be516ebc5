F='Wcomment\|^ *[0-9]* |\|^ *|\|warning generated\|In file included\|amdgpu.ids'
for v in 0 1; do
  export DDP_CONV32_PRIO=$v
  timeout 200 python tools/per_launch.py 2>&1 | grep -v "$F" | grep "conv32\|total" | tr '\n' ';'; echo " <- prio=$v"
done
