#!/usr/bin/env python3
"""Per-kernel means of raw rocprofv3 PMC counters (diagnostic passes, e.g. the SQ wave-state or TCP families).
Usage:  python tools/pmc_raw.py <rocprof -d dir> [kernel-name substring ...]  -> JSON on stdout"""
import glob
import json
import os
import sys

import pandas as pd

d = sys.argv[1]
want = sys.argv[2:] or ["ddp_conv_rows16_kernel", "ddp_conv_rows16_direct_kernel", "ddp_conv_rows_kernel", "ddp_conv32_kernel", "ddp_conv_messages_kernel", "ddp_stage_a_h2_kernel", "ddp_stage_a_mfma_kernel"]
f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
c = pd.read_csv(f)
out = {}
for k in want:
    s = c[c.Kernel_Name.str.contains(k, regex=False)]
    if not len(s):
        continue
    n = s.Dispatch_Id.nunique()
    row = {"launches_sampled": int(n)}
    for name, g in s.groupby("Counter_Name"):
        row[name] = float(g.Counter_Value.sum()) / n
    out[k] = row
print(json.dumps(out, indent=1))
