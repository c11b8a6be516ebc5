#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC passes (run on the GPU box):

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 bytes... (rocprofv3 reports kilobytes); on gfx950 FETCH_SIZE reads
exactly half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, section HBM), so it is doubled; WRITE_SIZE is
exact for 16-byte-per-lane streaming stores.  The result is averaged per ddp_conv_messages_kernel launch.
"""
import glob
import os
import json
import sys

import pandas as pd


KERNELS = {"ddp_conv32_kernel": "ddp_conv32_kernel", "ddp_conv_messages_kernel": "ddp_conv_messages_kernel"}


def per_launch(d, counter, kernel):
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)   # newest pass in the directory
    c = pd.read_csv(f)
    c = c[c.Kernel_Name.str.contains(kernel, regex=False) & (c.Counter_Name == counter)]
    per = c.groupby("Dispatch_Id").Counter_Value.sum()
    return float(per.mean()), int(per.shape[0])


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    res = {"corrections": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE x1; units of 1024 B", "kernels": {}}
    for name, pat in KERNELS.items():
        fetch_kb, n1 = per_launch(fetch_dir, "FETCH_SIZE", pat)
        write_kb, n2 = per_launch(write_dir, "WRITE_SIZE", pat)
        res["kernels"][name] = {"launches_sampled": [n1, n2], "FETCH_SIZE_kb_per_launch_raw": fetch_kb,
                                "WRITE_SIZE_kb_per_launch_raw": write_kb,
                                "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
    json.dump(res, open(out, "w"), indent=1)
    print(res)


if __name__ == "__main__":
    main()
