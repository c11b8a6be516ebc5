#!/usr/bin/env python3
"""Diagnostic: where does a conv workgroup spend its cycles?  Builds libddp_hip_stamps.so (-DDDP_STAMPS), runs a few
bench steps with it and prints the mean s_memtime deltas between the phase stamps of the LAST conv launch of layer 3
(the shape with all four weight blocks).  Usage on the GPU box:  python tools/stamp_conv.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffdock_pocket_amd import build  # noqa: E402

ABL = int(os.environ.get("DDP_STAMP_ABLATE", "0"))
os.environ["DDP_HIP_LIB"] = os.environ.get("DDP_STAMP_LIB") or build.build(stamps=True, ablate=ABL)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from diffdock_pocket_amd import _lib as L  # noqa: E402
from diffdock_pocket_amd import launch as sm  # noqa: E402  (engine.py launches through this module)
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

dev = torch.device("cuda:0")
model, kw = bench.build_model("cfg2", False, dev)
if "--direct" in sys.argv:
    model.factorize_min_degree = 0
g = make_3dpf_complex(seed=0, flexible_sidechains=False)
smp = Sampler(model, g, 40, dev, SamplerConfig(flexible_sidechains=False, hip_graph=False), seed=0)
smp.randomize()
lib = L.load()
lib.ddp_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
orig = sm.launch_convs
captured = {}


WANT_G = "--direct" not in sys.argv


def hooked(spec, tasks, **kw):
    kw.pop("tag", None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(spec, tasks, **kw)
    e1.record()
    torch.cuda.synchronize()
    sel = len(spec.blocks) == 4 and len(tasks) >= 5 and spec.factorized == WANT_G and spec.blocks[3].n == 60 and spec.blocks[2].n == 10 \
        and (spec.blocks[3].ntiles in (20, 140)) and (not WANT_G or min(spec.g_cols) > 0)
    if sel:
        captured["ms"] = e0.elapsed_time(e1)
    if sel and "done" not in captured:
        torch.cuda.synchronize()
        ET = 32 if spec.factorized else 64
        n = sum((t.n_edges + ET - 1) // ET for t in tasks)
        n = min(n, 32768)
        buf = np.zeros((n, 40), dtype=np.uint64)
        rc = lib.ddp_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), n)
        assert rc == 0
        captured["done"] = buf
        captured["tiles"] = [(t.n_edges + ET - 1) // ET for t in tasks]


sm.launch_convs = hooked
smp.step(0, get_t_schedule(20))
smp.step(1, get_t_schedule(20))
captured.pop("done", None)
smp.step(2, get_t_schedule(20))
st = captured["done"].astype(np.int64)
st = st[np.argsort(st[:, 37], kind="stable")]      # rows in tile order (workgroup ids are remapped per XCD)
captured["done"] = captured["done"][np.argsort(captured["done"][:, 37].astype(np.int64), kind="stable")]
W4 = True
if WANT_G and W4:   # ddp_conv32_kernel (4 waves): stamps 0..9 (see the kernel)
    names = ["stage edge_attr_", "fc1", "features (all blocks)", "wave0 role tiles", "wait other waves", "zero + park rounds",
             "G pass (wave 0)", "wait other waves (G)", "store rows"]
    idx = list(range(10))
elif WANT_G:        # ddp_conv32x6_kernel: 0 1 2 3 | 4 tile wave 0 done | 5 barrier | 6 parked | 9 stored; 7 = G wave 4 done
    names = ["stage edge_attr_", "fc1", "features (all blocks)", "wave0 role tiles", "wait other tile waves / G waves", "park rounds",
             "store rows"]
    idx = [0, 1, 2, 3, 4, 5, 6, 9]
else:
    names = ["stage edge_attr_", "fc1"]
    for b in range(4):
        names += [f"blk{b} features", f"blk{b} wave0 tiles", f"blk{b} wait other waves", f"blk{b} reduce+store"]
    idx = [0, 1, 2]
    for b in range(4):
        idx += [3 + 4 * b, 4 + 4 * b, 5 + 4 * b, 6 + 4 * b]
d = np.diff(st[:, idx], axis=1)
tot = (st[:, idx[-1]] - st[:, 0]).mean()
hw = captured["done"][:, 21]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
unit = xcc * 1000 + se * 100 + sh * 16 + cu
span = (st[:, 23].max() - st[:, 22].min()) / 100.0
busy = (st[:, 23] - st[:, 22]).sum() / 100.0
print(f"HIP-event time of this launch: {captured['ms']:.2f} ms  => s_memrealtime runs at {span * 100 / (captured['ms'] * 1e3):.1f} MHz")
print(f"launch span {span:.0f} us (if 100 MHz); distinct (xcc,se,sh,cu) units {len(np.unique(unit))}; distinct xcc {len(np.unique(xcc))}; "
      f"sum of workgroup times / span = {busy / span:.1f} concurrently resident workgroups")
rt = (st[:, 23] - st[:, 22]).mean()
print(f"in-kernel clock = {tot / rt * 100:.0f} MHz (s_memtime / s_memrealtime x 100 MHz); mean workgroup {rt / 100:.1f} us")
print(f"workgroups {len(st)}  mean total ticks {tot:.0f} (s_memtime ticks; 100 MHz constant clock on gfx950 => x24 for 2.4 GHz cycles)")
for n_, m in zip(names, d.mean(0)):
    print(f"  {n_:26s} {m:10.0f}  {100 * m / tot:5.1f} %")

if not WANT_G:
    sys.exit(0)
gph = (st[:, 7] - st[:, 6]).astype(np.float64) if W4 else (st[:, 7] - st[:, 3]).astype(np.float64)
print(f"G pass of the first G wave: {gph.mean():.0f} ticks")
tot_wg = (st[:, 9] - st[:, 0]).astype(np.float64)
start = (st[:, 22] - st[:, 22].min()) / 100.0
o = 0
print("per task (launch order): workgroups, mean G pass (wave 0), mean workgroup total, mean start time us")
for ti, nt in enumerate(captured["tiles"]):
    sl = slice(o, o + nt)
    print(f"  task {ti}: {nt:6d} wgs  G phase {gph[sl].mean():9.0f}  total {tot_wg[sl].mean():9.0f}  start {start[sl].mean():8.0f}")
    o += nt
order = np.argsort(st[:, 22])
dec = np.array_split(order, 10)
print("G phase by start-time decile:", " ".join(f"{gph[d].mean():.0f}" for d in dec))
print(f"workgroups {len(st)}")
gs = st[:, 24:32]
ns_ = st[:, 36]
print(f"g_stage wave 0 (slot 0): mean steps {ns_.mean():.1f}; cycles: prologue issue {(gs[:,1]-gs[:,0]).mean():.0f}, step0 {(gs[:,2]-gs[:,1]).mean():.0f}, "
      f"step1 {(gs[:,3]-gs[:,2]).mean():.0f}, step2 {(gs[:,4]-gs[:,3]).mean():.0f}, steps3-5 {(gs[:,5]-gs[:,4]).mean():.0f}, "
      f"rest of main loop {(gs[:,6]-gs[:,5]).mean():.0f}, extras {(gs[:,7]-gs[:,6]).mean():.0f}, total {(gs[:,7]-gs[:,0]).mean():.0f}")
o = 0
for ti, nt in enumerate(captured["tiles"]):
    sl = slice(o, o + nt)
    g = gs[sl]
    print(f"  task {ti}: steps {ns_[sl].mean():.1f} prologue {(g[:,1]-g[:,0]).mean():.0f} step0 {(g[:,2]-g[:,1]).mean():.0f} step1 {(g[:,3]-g[:,2]).mean():.0f} "
          f"step2 {(g[:,4]-g[:,3]).mean():.0f} steps3-5 {(g[:,5]-g[:,4]).mean():.0f} rest {(g[:,6]-g[:,5]).mean():.0f} extras {(g[:,7]-g[:,6]).mean():.0f}")
    o += nt
