"""Diagnostic: the kernel sequence of the LAST denoising step of a rocprofv3 kernel trace, run-length encoded, with start offsets
and durations in microseconds.  python tools/step_sequence.py <trace dir>"""
import glob
import sys

import pandas as pd

f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))
t = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
names = [n.replace("void at::native::", "").replace("(anonymous namespace)::", "")[:44] for n in t.Kernel_Name]
pose = [i for i, n in enumerate(names) if "ddp_pose_update" in n]
a, b = pose[-2] + 1, pose[-1] + 1
t0 = t.Start_Timestamp[a]
i = a
while i < b:
    j = i
    while j + 1 < b and names[j + 1] == names[i]:
        j += 1
    dur = sum(t.End_Timestamp[k] - t.Start_Timestamp[k] for k in range(i, j + 1)) / 1e3
    print(f"{(t.Start_Timestamp[i] - t0) / 1e3:9.1f} us  x{j - i + 1:<3d} {dur:8.1f} us  {names[i]}")
    i = j + 1
print("step span", (t.End_Timestamp[b - 1] - t0) / 1e3, "us; kernels", b - a)
