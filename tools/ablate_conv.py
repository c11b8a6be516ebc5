#!/usr/bin/env python3
"""Diagnostic: HIP-event time of one layer-3 conv launch (9 convs, 40 samples) when only a subset of the four weight
blocks is executed (outputs are then incomplete - timing only).  python tools/ablate_conv.py"""
import copy
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
orig = sm._launch_convs
saved = {}


def hooked(spec, tasks):
    if len(spec.blocks) == 4 and len(tasks) == 9 and "t" not in saved:
        saved["t"] = (spec, list(tasks))
    orig(spec, tasks)


sm._launch_convs = hooked
smp.step(0, get_t_schedule(20))
torch.cuda.synchronize()
spec, tasks = saved["t"]
edges = sum(t.n_edges for t in tasks)
names = ["0e", "1o", "1e", "0o"]


def timeit(blocks, reps=3):
    sp = copy.copy(spec)
    sp.blocks = [spec.blocks[i] for i in blocks]
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(sp, tasks)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print(f"edges {edges}, workgroups {sum((t.n_edges + 63) // 64 for t in tasks)}")
base = None
for sel in ([0, 1, 2, 3], [0], [3], [1], [2], [0, 3], [1, 2], [3, 0]):
    ms = timeit(sel)
    tiles = sum(spec.blocks[i].ntiles for i in sel)
    mf = edges / 64 * (tiles * 2 * (spec.hp // 8) * 4 + spec.nct1 * 2 * (spec.kp1 // 8) * 4) * 4096 / (ms * 1e-3) / 1e12
    print(f"blocks {[names[i] for i in sel]!s:28s} {ms:8.2f} ms   executed MFMA rate {mf:6.1f} TFLOP/s")

# --- does the time follow the weights (tile0) or the block's features?
import dataclasses
b0, b3 = spec.blocks[0], spec.blocks[3]
sw0 = dataclasses.replace(b0, segs=list(b0.segs)); sw0.tile0, sw0.ntiles, sw0.nsub, sw0.ups = b3.tile0, b3.ntiles, b0.nsub, b0.ups
sw3 = dataclasses.replace(b3, segs=list(b3.segs)); sw3.tile0, sw3.ntiles, sw3.nsub, sw3.ups = b0.tile0, b0.ntiles, b3.nsub, b3.ups
for name, blk in (("0e features + 0o weights", sw0), ("0o features + 0e weights", sw3)):
    sp = copy.copy(spec)
    sp.blocks = [blk]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    orig(sp, tasks); torch.cuda.synchronize()
    e0.record(); orig(sp, tasks); e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1):8.2f} ms")
# --- per task
for i, t in enumerate(tasks):
    for sel in ([0], [3]):
        sp = copy.copy(spec); sp.blocks = [spec.blocks[j] for j in sel]
        orig(sp, [t]); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(sp, [t]); e1.record(); torch.cuda.synchronize()
        print(f"task {i} edges {t.n_edges:7d} block {names[sel[0]]}: {e0.elapsed_time(e1):7.2f} ms")

# --- value or address?  overwrite the 0e tile region of conv 3*9+3 (atom-atom) with the 0o tile data / random / zeros
conv = model.conv_layers[9 * 3 + 3]
w = conv._packed.w2p
tile_f = spec.hp * 32
t3 = tasks[3]
def t_block(sel):
    sp = copy.copy(spec); sp.blocks = [spec.blocks[j] for j in sel]
    orig(sp, [t3]); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(sp, [t3]); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
print("w2p numel", w.numel(), "expected", spec.ntiles * tile_f, "abs mean 0e/0o regions", float(w[:140 * tile_f].abs().mean()), float(w[194 * tile_f:334 * tile_f].abs().mean()))
print("aa conv: 0e %.2f ms, 0o %.2f ms" % (t_block([0]), t_block([3])))
keep = w[:140 * tile_f].clone()
w[:140 * tile_f] = w[194 * tile_f:334 * tile_f]
print("0e region <- 0o data: 0e %.2f ms" % t_block([0]))
w[:140 * tile_f] = torch.randn_like(keep) * 0.05
print("0e region <- randn*0.05: 0e %.2f ms" % t_block([0]))
w[:140 * tile_f] = 0
print("0e region <- zeros: 0e %.2f ms" % t_block([0]))
w[:140 * tile_f] = keep
w[194 * tile_f:334 * tile_f] = keep
print("0o region <- 0e data: 0o %.2f ms" % t_block([3]))

# --- time vs tile0 / ntiles for the aa conv (0e features)
for tile0, nt in ((0, 140), (20, 140), (60, 140), (100, 140), (140, 140), (180, 140), (194, 140), (0, 70), (70, 70), (194, 70), (264, 70), (0, 16), (318, 16)):
    blk = dataclasses.replace(b0, segs=list(b0.segs)); blk.tile0, blk.ntiles = tile0, nt
    sp = copy.copy(spec); sp.blocks = [blk]
    orig(sp, [t3]); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(sp, [t3]); e1.record(); torch.cuda.synchronize()
    print(f"tile0 {tile0:4d} ntiles {nt:4d}: {e0.elapsed_time(e1):6.2f} ms")
