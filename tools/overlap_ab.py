"""Round 6, VERDICT item 1: can stage A (HBM-bound) and ddp_conv_rows (matrix-bound) share CUs?  Four measurements on the REAL launches
of one 40-sample cfg2 step (layer 3: the largest), taken inside the forward while every buffer of the launches is alive:

  rows alone      the layer's factorised-conv launch at two workgroups per CU (the kernel's own occupancy) and at ONE (dynamic LDS >= 82 KiB
                  through ddp_set_occupancy_shaping: one 256-register wave per SIMD, the other wave slot and 78 KiB of LDS left free)
  stage A alone   the atom rows' product of the same layer (ddp_stage_a_gh, 4.7 GB of G) at two workgroups per CU and at one
  the pair        both launches on two streams, unshaped and shaped (rows at one workgroup per CU), wall time from the first start to the last end
  the step        bench-style: captured steps with model.shape_early_rows on / off (the early conv launch of a layer at one workgroup per CU
                  beside stage A of the atom rows)

Prints one text block (profiles/r06_overlap_ab.txt).  Diagnostic; results of the launches do not depend on the shaping (same kernels, same
arguments)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from diffdock_pocket_amd import _lib as L  # noqa: E402
from diffdock_pocket_amd import launch as K  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

ROWS_ONE = 82 * 1024
SA_ONE = 30 * 1024
REPS = 6


class _PairsDone(Exception):
    pass


def shaping(rows=0, sa=0):
    L.check(L.load().ddp_set_occupancy_shaping(rows, sa), "shaping")


def ev():
    return torch.cuda.Event(enable_timing=True)


def time_one(fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = ev(), ev()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def time_pair(fa, fb, sa, sb, reps=REPS):
    """fa on stream sa, fb on stream sb, both started behind one event; returns (wall, fa's time, fb's time), medians."""
    out = []
    for i in range(reps + 1):
        torch.cuda.synchronize()
        start = ev()
        start.record()
        ea0, ea1, eb0, eb1 = ev(), ev(), ev(), ev()
        with torch.cuda.stream(sa):
            sa.wait_event(start)
            ea0.record()
            fa()
            ea1.record()
        with torch.cuda.stream(sb):
            sb.wait_event(start)
            eb0.record()
            fb()
            eb1.record()
        torch.cuda.synchronize()
        if i:
            out.append((max(start.elapsed_time(ea1), start.elapsed_time(eb1)), ea0.elapsed_time(ea1), eb0.elapsed_time(eb1)))
    out.sort()
    return out[len(out) // 2]


def main():
    dev = torch.device("cuda:0")
    model, _ = bench.build_model("cfg2", False, dev)
    cg = make_3dpf_complex(seed=0, flexible_sidechains=False)
    smp = Sampler(model, cg, 40, dev, SamplerConfig(inference_steps=20, flexible_sidechains=False, hip_graph=False), seed=0)
    smp.randomize()
    sched = get_t_schedule(20)
    model.split_rows_launch = False          # one conv launch per layer: the launch IS the layer's factorised convs
    for i in range(3):
        smp.step(i, sched)
    torch.cuda.synchronize()
    res = {}
    last_sa = {}
    real_stage_a, real_launch = K.stage_a, K.launch_convs
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def stage_a(x, n_rows, offs, nb, W, out, **kw):
        real_stage_a(x, n_rows, offs, nb, W, out, **kw)
        if kw.get("gh") is not None and kw.get("rows") is None and n_rows >= 40000:
            last_sa["call"] = lambda: real_stage_a(x, n_rows, offs, nb, W, out, **kw)
            last_sa["gb"] = out.numel() * 4 / 1e9

    def launch_convs(spec, tasks, flops_spec=None, node_bytes=0.0, tag=None):
        real_launch(spec, tasks, flops_spec=flops_spec, node_bytes=node_bytes, tag=tag)
        if tag != "layer3" or not all(getattr(t, "_rows", False) for t in tasks) or "rows2" in res or "call" not in last_sa:
            return
        torch.cuda.synchronize()
        rows = lambda: real_launch(spec, tasks, flops_spec=flops_spec, node_bytes=node_bytes, tag=tag)      # noqa: E731
        sa = last_sa["call"]
        if "--pair-only" in sys.argv:      # under rocprofv3 --kernel-trace: one unshaped and one shaped pair, marked by a tiny fill in front of each
            res["rows2"] = 0.0
            for shaped in (0, ROWS_ONE):
                torch.cuda.synchronize()
                torch.zeros(1, device=dev).fill_(1.0)
                shaping(shaped, 0)
                time_pair(rows, sa, s1, s2, reps=1)
            shaping(0, 0)
            raise _PairsDone()      # (the pairs stay the LAST dispatches of the trace: tools/pair_offsets.py reads them from its end)
        res["edges"] = sum(t.n_edges for t in tasks)
        res["sa_gb"] = last_sa["gb"]
        shaping(0, 0)
        res["rows2"] = time_one(rows)
        res["sa2"] = time_one(sa)
        shaping(ROWS_ONE, SA_ONE)
        res["rows1"] = time_one(rows)
        res["sa1"] = time_one(sa)
        shaping(0, 0)
        res["pair_unshaped"] = time_pair(rows, sa, s1, s2)
        res["pair_unshaped_sa_first"] = time_pair(sa, rows, s2, s1)
        shaping(ROWS_ONE, 0)
        res["pair_rows1"] = time_pair(rows, sa, s1, s2)
        res["pair_rows1_sa_first"] = time_pair(sa, rows, s2, s1)
        shaping(ROWS_ONE, SA_ONE)
        res["pair_both1"] = time_pair(rows, sa, s1, s2)
        shaping(0, 0)

    K.stage_a, K.launch_convs = stage_a, launch_convs
    try:
        smp.step(3, sched)
        torch.cuda.synchronize()
    except _PairsDone:
        torch.cuda.synchronize()
    finally:
        K.stage_a, K.launch_convs = real_stage_a, real_launch
        shaping(0, 0)
    smp.close()
    if "--pair-only" in sys.argv:
        return
    print(f"# layer 3 of a 40-sample cfg2 step (schedule position 3): {res['edges']} edges in the conv launch, {res['sa_gb']:.2f} GB of G from the atom rows' stage A")
    print(f"rows alone      two workgroups per CU {res['rows2']:.3f} ms | one per CU (dynamic LDS {ROWS_ONE // 1024} KiB) {res['rows1']:.3f} ms = x{res['rows1'] / res['rows2']:.2f}")
    print(f"stage A alone   two workgroups per CU {res['sa2']:.3f} ms ({res['sa_gb'] / res['sa2']:.2f} TB/s) | one per CU (+{SA_ONE // 1024} KiB) {res['sa1']:.3f} ms "
          f"({res['sa_gb'] / res['sa1']:.2f} TB/s) = x{res['sa1'] / res['sa2']:.2f}")
    ser = res["rows2"] + res["sa2"]
    for key, what in (("pair_unshaped", "unshaped, rows queued first"), ("pair_unshaped_sa_first", "unshaped, stage A queued first"),
                      ("pair_rows1", "rows at one per CU, rows queued first"), ("pair_rows1_sa_first", "rows at one per CU, stage A queued first"),
                      ("pair_both1", "both at one per CU")):
        w, a, b = res[key]
        first, second = ("rows", "stage A") if not key.endswith("sa_first") else ("stage A", "rows")
        print(f"the pair        {what}: wall {w:.3f} ms ({first} {a:.3f}, {second} {b:.3f}) against {ser:.3f} one after the other = x{w / ser:.2f}")

    # ---- the step: captured, the early conv launch shaped or not
    for shaped in (False, True, False, True):
        m2, _ = bench.build_model("cfg2", False, dev)
        m2.shape_early_rows = shaped
        s_ = Sampler(m2, cg, 40, dev, SamplerConfig(inference_steps=20, flexible_sidechains=False), seed=0)
        s_.randomize()
        for i in range(4):
            s_.step(i, sched)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            s_.step(i, sched)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        s_.check_overflow()
        print(f"the step        shape_early_rows={shaped}: {ms:.3f} ms per step (20 replayed steps)")
        s_.close()
        del s_, m2


if __name__ == "__main__":
    main()
