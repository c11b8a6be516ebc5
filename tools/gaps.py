#!/usr/bin/env python3
"""Device idle time per denoising step from a rocprofv3 kernel trace (diagnostic).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline
    python tools/gaps.py gpurun_out/trace

Kernels of all streams are merged into busy intervals; a step = the span between two consecutive ddp_pose_update launches.
Prints busy / idle per step and where the idle time sits (the kernel that ends each gap > 20 us)."""
import glob
import sys
from collections import Counter

import pandas as pd

f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))
t = pd.read_csv(f).sort_values("Start_Timestamp")
names, st, en = t.Kernel_Name.tolist(), t.Start_Timestamp.tolist(), t.End_Timestamp.tolist()
pose = [i for i, n in enumerate(names) if "ddp_pose_update" in n]
rows, after = [], Counter()
for a, b in zip(pose[2:-1], pose[3:]):          # skip warm-up steps
    busy, cur_end, idle = 0, st[a], 0
    for i in range(a, b):
        if st[i] > cur_end:
            gap = st[i] - cur_end
            idle += gap
            if gap > 20000:
                after[names[i][:60]] += gap
            cur_end = st[i]
        if en[i] > cur_end:
            busy += en[i] - max(st[i], cur_end)
            cur_end = en[i]
    rows.append((st[b] - st[a], busy, idle))
n = len(rows)
print(f"{n} steps: span {sum(r[0] for r in rows) / n / 1e6:.2f} ms, busy {sum(r[1] for r in rows) / n / 1e6:.2f} ms, "
      f"idle {sum(r[2] for r in rows) / n / 1e6:.2f} ms per step")
print("idle time in gaps > 20 us, by the kernel that ends the gap (ms per step):")
for k, v in after.most_common(15):
    print(f"  {v / n / 1e6:7.3f}  {k}")
