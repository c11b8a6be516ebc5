"""Diagnostic: which kernels run right before / after every runtime buffer copy (__amd_rocclr_copyBuffer) of a rocprofv3 kernel
trace - i.e. which host call issues it.  python tools/copy_neighbours.py <trace dir>"""
import glob
import sys
from collections import Counter

import pandas as pd

f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))
t = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
names = [n.replace("void at::native::", "")[:50] for n in t.Kernel_Name]
pairs = Counter()
for i, n in enumerate(names):
    if "copyBuffer" in n or "fillBuffer" in n:
        pairs[(n[:22], names[i - 1] if i else "-", names[i + 1] if i + 1 < len(names) else "-")] += 1
for (n, a, b), c in pairs.most_common(40):
    print(f"{c:5d}  {n:22s} after [{a}]  before [{b}]")
