#!/usr/bin/env python3
"""Diagnostic: per factorised conv task of one denoising step, how many G units a 32-edge tile has under different unit
rules: runs of one source node cut at 8 / 16 / 32 edges.  (A unit = one pass over the node's 100 kB G row.)
Usage on the GPU box:  python tools/unit_stats.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from diffdock_pocket_amd import score_model as sm  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

dev = torch.device("cuda:0")
model, kw = bench.build_model("cfg2", False, dev)
g = make_3dpf_complex(seed=0, flexible_sidechains=False)
smp = Sampler(model, g, 40, dev, SamplerConfig(flexible_sidechains=False), seed=0)
smp.randomize()
sched = get_t_schedule(20)
smp.step(0, sched)
seen = []
orig = sm._make_task


def hooked(pk, x_src, ldx_src, csr, sh, segs, msg, g=None, pos=None):
    if g is not None and csr.n_edges > 0:      # factorised task: `csr` is the source-ordered view
        seen.append(csr)
    return orig(pk, x_src, ldx_src, csr, sh, segs, msg, g=g, pos=pos)


sm._make_task = hooked
smp.step(1, sched)
torch.cuda.synchronize()
tot = {8: 0, 16: 0, 32: 0}
tiles_all = 0
print(f"{'edges':>9s} {'tiles':>7s} {'deg':>6s} {'runs/tile':>9s}  units/tile at cut 8 / 16 / 32")
for so in seen:
    src = so.src.long()[: so.n_edges]
    E = src.shape[0]
    tile = torch.arange(E, device=dev) // 32
    new = torch.ones(E, dtype=torch.bool, device=dev)
    new[1:] = (src[1:] != src[:-1]) | (tile[1:] != tile[:-1])
    run_id = torch.cumsum(new, 0) - 1
    lens = torch.bincount(run_id)
    nt = int(tile[-1]) + 1
    deg = E / int(torch.unique(src).numel())
    row = [float(((lens + c - 1) // c).sum()) / nt for c in (8, 16, 32)]
    for c, r in zip((8, 16, 32), row):
        tot[c] += r * nt
    tiles_all += nt
    print(f"{E:9d} {nt:7d} {deg:6.1f} {float(lens.numel()) / nt:9.2f}  {row[0]:.2f} / {row[1]:.2f} / {row[2]:.2f}")
print("all tasks, units per tile:", {c: round(v / tiles_all, 2) for c, v in tot.items()})
