"""Start / end offsets of the kernels of tools/overlap_ab.py --pair-only from a rocprofv3 --kernel-trace directory: for the last two pairs
(unshaped, rows shaped to one workgroup per CU) the dispatches of ddp_conv_rows and ddp_stage_a_h2 with their times relative to the pair's
first start - are the two kernels resident together?"""
import glob
import os
import sys

import pandas as pd

d = sys.argv[1]
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
t = pd.read_csv(f).sort_values("Start_Timestamp")
print("# columns:", ", ".join(t.columns))
t = t[t.Kernel_Name.str.contains("ddp_conv_rows|ddp_stage_a_h2")]
# the pairs are the last dispatches of the trace: 2 pairs x (1 warm-up + 1 timed) x 2 kernels
last = t.tail(8)
for name, grp in (("unshaped (two rows workgroups per CU)", last.iloc[2:4]), ("rows at one workgroup per CU", last.iloc[6:8])):
    t0 = grp.Start_Timestamp.min()
    print(name)
    for _, r in grp.iterrows():
        print(f"   {r.Kernel_Name[:34]:34s} start {(r.Start_Timestamp - t0) / 1e3:9.1f} us  end {(r.End_Timestamp - t0) / 1e3:9.1f} us  "
              f"LDS {r.get('LDS_Block_Size', 0)} B  arch VGPRs {r.get('VGPR_Count', 0)} + acc {r.get('Accum_VGPR_Count', 0)}  workgroups {int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1))), 1)}")
    a, b = grp.iloc[0], grp.iloc[1]
    ov = min(a.End_Timestamp, b.End_Timestamp) - max(a.Start_Timestamp, b.Start_Timestamp)
    print(f"   both in flight for {ov / 1e3:.1f} us of {(grp.End_Timestamp.max() - t0) / 1e3:.1f} us")
