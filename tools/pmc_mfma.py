#!/usr/bin/env python3
"""Matrix-pipe utilisation of the conv kernels from a rocprofv3 PMC pass (run on the GPU box):

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \\
            --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-pass
  python tools/pmc_mfma.py gpurun_out/pmc_mfma profiles/r01_mfma_pmc.json

Per kernel instantiation and launch: SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's MFMA unit is busy, summed over SIMDs),
GRBM_GUI_ACTIVE (cycles the dispatch keeps the GPU active), SQ_INSTS_VALU_MFMA_MOPS_F32 (fp32 MFMA work issued).
mfma_busy_frac = MFMA_BUSY / (GUI_ACTIVE_per_xcd * 1024 SIMDs) with GUI_ACTIVE summed over the 8 XCDs, i.e. the gfx94x
`MfmaUtil` formula (ROCm 7.2 has no gfx950 section for derived counters, MI355X_MICROARCH.md)."""
import glob
import json
import os
import sys

import pandas as pd

KERNELS = ["ddp_conv32_kernel", "ddp_conv_messages_kernel", "ddp_stage_a_mfma_kernel"]


def main():
    d, out = sys.argv[1:3]
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)
    c = pd.read_csv(f)
    res = {"formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)", "kernels": {}}
    for k in KERNELS:
        ck = c[c.Kernel_Name.str.contains(k, regex=False)]
        if ck.empty:
            continue
        per = ck.pivot_table(index="Dispatch_Id", columns="Counter_Name", values="Counter_Value", aggfunc="sum")
        m = per.mean()
        gui = float(m.get("GRBM_GUI_ACTIVE", float("nan")))
        busy = float(m.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")))
        res["kernels"][k] = {"launches_sampled": int(per.shape[0]), **{n: float(v) for n, v in m.items()},
                             "mfma_busy_frac": busy / (gui / 8.0 * 1024.0) if gui > 0 else None}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
