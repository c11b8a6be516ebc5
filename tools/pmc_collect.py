#!/usr/bin/env python3
"""Consolidate the rocprofv3 PMC passes of one build into profiles/r05_pmc.json (read by bench.py).

Run on the GPU box, every pass in its own process with --kernel-trace only (gpurun refuses --pmc with other trace domains),
each on the same deterministic workload:

  B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-pass --no-other-workloads"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- $B --launch-log gpurun_out/pmc_fetch/launches.json
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- $B --launch-log gpurun_out/pmc_write/launches.json
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \\
            -d gpurun_out/pmc_mfma -- $B --launch-log gpurun_out/pmc_mfma/launches.json
  python3 tools/pmc_collect.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma profiles/r05_pmc.json

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in units of 1024 bytes; on gfx950
FETCH_SIZE counts a 128-byte request as 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
One MOP of SQ_INSTS_VALU_MFMA_MOPS_F32 / _F16 = 512 FLOP.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x
1024 SIMDs).

Launch matching: bench.py --launch-log lists the conv launches of the TIMED steps in order (kernel instantiation, edges,
useful FLOPs); with --no-hbm-pass / --no-other-workloads these are the last conv dispatches of the process, so the last
len(log) dispatches of each instantiation are the logged ones.  padding_frac = 1 - sum(useful) / sum(issued) over exactly
those launches.  Per step: HBM bytes of ALL dispatches between the first and the last logged conv dispatch / timed steps."""
import glob
import json
import os
import sys

import pandas as pd

CONV = ["ddp_conv_rows16_kernel", "ddp_conv_rows16_direct_kernel", "ddp_conv_rows_kernel", "ddp_conv32_kernel", "ddp_conv_messages_kernel"]
OTHER = ["ddp_stage_a_h2_kernel", "ddp_stage_a_mfma_kernel", "ddp_segment_reduce4_kernel", "ddp_segment_reduce_kernel", "ddp_edge_featurize", "ddp_radius", "ddp_knn", "ddp_pose_update"]


def load(d):
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    c = pd.read_csv(f)
    with open(os.path.join(d, "launches.json")) as fh:
        log = json.load(fh)
    return c, log


def timed_window(c, log):
    """Dispatch ids of the logged (timed) conv launches per instantiation + the [first, last] dispatch-id window."""
    ids = {}
    for k in CONV:
        n = sum(1 for l in log["launches"] if l["kernel"] == k)
        d = sorted(c[c.Kernel_Name.str.contains(k, regex=False)].Dispatch_Id.unique())
        if n and len(d) >= n:
            ids[k] = d[-n:]
    lo = min(v[0] for v in ids.values())
    hi = max(v[-1] for v in ids.values())
    return ids, lo, hi


def counter_sum(c, name, disp_ids=None, lo=None, hi=None, kernel=None):
    s = c[c.Counter_Name == name]
    if kernel is not None:
        s = s[s.Kernel_Name.str.contains(kernel, regex=False)]
    if disp_ids is not None:
        s = s[s.Dispatch_Id.isin(disp_ids)]
    if lo is not None:
        s = s[(s.Dispatch_Id >= lo) & (s.Dispatch_Id <= hi)]
    return float(s.Counter_Value.sum()), int(s.Dispatch_Id.nunique())


def l2_block(l2_dir, lat_dir):
    """Per-kernel L2 figures from two further passes (TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum ... / TCP_TCC_READ_REQ_LATENCY_sum
    TCC_BUSY_avr TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE): requests are 128 bytes; GRBM_GUI_ACTIVE sums the 8 XCDs."""
    out = {}
    def per_kernel(d):
        f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
        c = pd.read_csv(f)
        res = {}
        for k in CONV + OTHER:
            s = c[c.Kernel_Name.str.contains(k, regex=False)]
            if len(s):
                n = s.Dispatch_Id.nunique()
                res[k] = {name: float(g.Counter_Value.sum()) / n for name, g in s.groupby("Counter_Name")}
                res[k]["launches_sampled"] = int(n)
        return res
    a, b = per_kernel(l2_dir), per_kernel(lat_dir)
    for k in a:
        r, t = a[k], b.get(k, {})
        gui = r.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        e = {"launches_sampled": r["launches_sampled"], "l2_requests_per_launch": r.get("TCC_REQ_sum"), "l2_request_bytes_per_launch": 128.0 * r.get("TCC_REQ_sum", 0.0),
             "l2_hit_rate": r.get("TCC_HIT_sum", 0.0) / max(r.get("TCC_HIT_sum", 0.0) + r.get("TCC_MISS_sum", 0.0), 1.0),
             "cycles_per_launch": gui, "l2_bytes_per_cycle_chip": 128.0 * r.get("TCC_REQ_sum", 0.0) / gui if gui else None,
             "ta_busy_frac": r.get("TA_BUSY_avr", 0.0) / gui if gui else None,
             "tcp_pending_stall_frac": r.get("TCP_PENDING_STALL_CYCLES_sum", 0.0) / (gui * 256.0) if gui else None}
        if t:
            gui2 = t.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            e["l2_busy_frac"] = t.get("TCC_BUSY_avr", 0.0) / gui2 if gui2 else None
            e["tcp_tcc_read_latency_cycles"] = t.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / max(t.get("TCP_TCC_READ_REQ_sum", 1.0), 1.0)
            e["hbm_read_bytes_per_launch_ea"] = 128.0 * t.get("TCC_EA0_RDREQ_sum", 0.0)
        out[k] = e
    return out


def main():
    fetch_dir, write_dir, mfma_dir, out = sys.argv[1:5]
    (cf, lf), (cw, lw), (cm, lm) = load(fetch_dir), load(write_dir), load(mfma_dir)
    assert lf["src_sha16"] == lw["src_sha16"] == lm["src_sha16"] and lf["workload"] == lw["workload"] == lm["workload"]
    steps = None
    res = {"src_sha16": lf["src_sha16"], "workload": lf["workload"],
           "corrections": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE x1, units of 1024 B; 1 MFMA MOP = 512 FLOP",
           "commands": __doc__.split("Corrections")[0].strip().splitlines()[3:],
           "kernels": {}}
    idf, lof, hif = timed_window(cf, lf)
    idw, low, hiw = timed_window(cw, lw)
    idm, lom, him = timed_window(cm, lm)
    for k in CONV:
        if k not in idf:
            continue
        n = len(idf[k])
        fe, _ = counter_sum(cf, "FETCH_SIZE", idf[k])
        wr, _ = counter_sum(cw, "WRITE_SIZE", idw[k])
        mops, _ = counter_sum(cm, "SQ_INSTS_VALU_MFMA_MOPS_F32", idm[k])
        mops16, _ = counter_sum(cm, "SQ_INSTS_VALU_MFMA_MOPS_F16", idm[k])
        busy, _ = counter_sum(cm, "SQ_VALU_MFMA_BUSY_CYCLES", idm[k])
        gui, _ = counter_sum(cm, "GRBM_GUI_ACTIVE", idm[k])
        useful = sum(l["useful_flops"] for l in lm["launches"] if l["kernel"] == k)
        # matrix-core instruction FLOPs of the kernel's formulation without padding: the fc products that ran as fp16 hi/lo split
        # products count three instruction FLOPs per product FLOP (bench.py --launch-log: fc16_flops), the rest once
        fc16 = sum(l.get("fc16_flops", 0.0) for l in lm["launches"] if l["kernel"] == k)
        useful_instr = useful + 2.0 * fc16
        issued = (mops + mops16) * 512.0
        res["kernels"][k] = {"launches_sampled": n, "hbm_bytes_per_launch": (2.0 * fe + wr) * 1024.0 / n,
                             "fetch_bytes_per_launch": 2.0 * fe * 1024.0 / n, "write_bytes_per_launch": wr * 1024.0 / n,
                             "issued_mfma_gflop_per_launch": issued / n / 1e9, "issued_f16_mfma_gflop_per_launch": mops16 * 512.0 / n / 1e9,
                             "issued_f32_mfma_gflop_per_launch": mops * 512.0 / n / 1e9,
                             "useful_gflop_per_launch_same_launches": useful / n / 1e9,
                             "useful_instruction_gflop_per_launch_same_launches": useful_instr / n / 1e9,
                             "padding_frac": 1.0 - useful_instr / issued if issued > 0 else None,
                             "mfma_busy_frac": busy / (gui / 8.0 * 1024.0) if gui > 0 else None}
    # the other kernels inside the timed window, per launch
    for k in OTHER:
        fe, n1 = counter_sum(cf, "FETCH_SIZE", lo=lof, hi=hif, kernel=k)
        wr, n2 = counter_sum(cw, "WRITE_SIZE", lo=low, hi=hiw, kernel=k)
        if n1 == 0 or n2 == 0:
            continue
        busy, _ = counter_sum(cm, "SQ_VALU_MFMA_BUSY_CYCLES", lo=lom, hi=him, kernel=k)
        gui, _ = counter_sum(cm, "GRBM_GUI_ACTIVE", lo=lom, hi=him, kernel=k)
        res["kernels"][k] = {"launches_sampled": n1, "hbm_bytes_per_launch": 2.0 * fe * 1024.0 / n1 + wr * 1024.0 / n2,
                             "fetch_bytes_per_launch": 2.0 * fe * 1024.0 / n1, "write_bytes_per_launch": wr * 1024.0 / n2,
                             "mfma_busy_frac": busy / (gui / 8.0 * 1024.0) if gui > 0 else None}
    # whole step
    dom = max(idf, key=lambda k: len(idf[k]))
    per_step_launches = {}
    for k in idf:
        per_step_launches[k] = len(idf[k])
    fe_all, _ = counter_sum(cf, "FETCH_SIZE", lo=lof, hi=hif)
    wr_all, _ = counter_sum(cw, "WRITE_SIZE", lo=low, hi=hiw)
    # timed steps = logged launches of the dominant instantiation / its launches per step (bench.py logs `steps` in the workload run)
    steps = int(lf.get("steps", 0)) or None
    if steps is None:   # infer: ddp_pose_update runs once per step
        pose = cf[cf.Kernel_Name.str.contains("ddp_pose_update", regex=False) & (cf.Dispatch_Id >= lof) & (cf.Dispatch_Id <= hif)]
        steps = max(int(pose.Dispatch_Id.nunique()), 1)
    res["_per_step"] = {"timed_steps": steps, "hbm_bytes_per_step": (2.0 * fe_all + wr_all) * 1024.0 / steps,
                        "fetch_bytes_per_step": 2.0 * fe_all * 1024.0 / steps, "write_bytes_per_step": wr_all * 1024.0 / steps,
                        "window": "all dispatches from the first to the last logged conv launch (the last step's pose update falls outside)"}
    if len(sys.argv) >= 7:      # optional: the L1 / L2 passes
        # those passes carry no launch log and count every step of their run (warm-up + timed, the SPLIT launches of the timed region's
        # layer order) while the main passes sample the instrumented pass (one launch per layer): every figure is therefore also given per
        # STEP and per whole-layer launch of the main passes (`*_per_layer_launch`: the basis of `avg_launch_ms` in bench.py's line)
        pass_steps = int(lf.get("warmup_steps", 0)) + int(lf.get("steps", 0))
        for k, e in l2_block(sys.argv[5], sys.argv[6]).items():
            main_per_step = (res["kernels"].get(k, {}).get("launches_sampled") or 0) / steps if steps else 0
            if pass_steps and e.get("launches_sampled"):
                lps = e["launches_sampled"] / pass_steps
                e["steps_in_pass"] = pass_steps
                e["launches_per_step"] = lps
                e["l2_request_bytes_per_step"] = e["l2_request_bytes_per_launch"] * lps
                if e.get("hbm_read_bytes_per_launch_ea") is not None:
                    e["hbm_read_bytes_per_step_ea"] = e["hbm_read_bytes_per_launch_ea"] * lps
                if main_per_step:
                    e["main_pass_launches_per_step"] = main_per_step
                    e["l2_request_bytes_per_layer_launch"] = e["l2_request_bytes_per_step"] / main_per_step
            res["kernels"].setdefault(k, {})["l2"] = e
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
