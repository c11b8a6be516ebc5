#!/usr/bin/env python3
"""Diagnostic: where does a workgroup of ddp_conv_rows_kernel spend its cycles?  Loads libddp_hip_rowstamps.so (-DDDP_ROWS_STAMPS; build it
before gpurun: python -c "from diffdock_pocket_amd import build; build.build(defs=['DDP_ROWS_STAMPS'], tag='rowstamps')"), runs three
launch-by-launch steps of the 40-sample job and prints the s_memtime deltas between the phase stamps of the layer-3 launch of the last
step: per phase the mean over workgroups of (mean over the 4 waves, slowest wave).   python tools/stamp_rows.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DDP_HIP_LIB"] = os.environ.get("STAMP_LIB", os.path.join(ROOT, "diffdock_pocket_amd", "libddp_hip_rowstamps.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from diffdock_pocket_amd import _lib as L  # noqa: E402
from diffdock_pocket_amd import launch as sm  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

dev = torch.device("cuda:0")
model, kw = bench.build_model("cfg2", False, dev)
model.overlap_direct_conv = False
model.rows_mfma16 = False      # (the stamps sit in ddp_conv_rows.hip, the round-5 kernel on v_mfma_f32_32x32x16_f16; the 16x16x32 kernel has none)
g = make_3dpf_complex(seed=0, flexible_sidechains=False)
smp = Sampler(model, g, 40, dev, SamplerConfig(flexible_sidechains=False, hip_graph=False), seed=0)
smp.randomize()
lib = L.load()
lib.ddp_debug_read_rows_stamps.argtypes = [C.c_void_p, C.c_int]
orig = sm.launch_convs
captured = {}
WANT = int(os.environ.get("STAMP_LAYER", "3"))


def hooked(spec, tasks, **kw):
    tag = kw.get("tag")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(spec, tasks, **kw)
    e1.record()
    if spec.factorized and tag == f"layer{WANT}" and all(getattr(t, "_rows", False) for t in tasks):
        torch.cuda.synchronize()
        n = min(sum((t.n_edges + 127) // 128 for t in tasks), 16384)
        buf = np.zeros((n, 4, 32), dtype=np.uint64)
        assert lib.ddp_debug_read_rows_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
        captured["buf"], captured["ms"], captured["wgs"] = buf, e0.elapsed_time(e1), sum((t.n_edges + 127) // 128 for t in tasks)


sm.launch_convs = hooked
sched = get_t_schedule(20)
for i in range(3):
    smp.step(i, sched)
st = captured["buf"].astype(np.int64)
print(f"layer {WANT} launch: {captured['ms']:.3f} ms, {captured['wgs']} workgroups (stamps of the first {st.shape[0]})")
ok = st[:, :, 0].min(axis=1) > 0
st = st[ok]
names = ["indices + stage tile 0", "edge_attr_ gather + split", "fc1 (6 tiles)", "per-edge tables"]
idx = [0, 1, 2, 3, 4]
for s in range(6):
    names += [f"seg{s} features + G runs", f"seg{s} stream tiles", f"seg{s} lane sum + store"]
    idx += [5 + 3 * s, 6 + 3 * s, 7 + 3 * s]
tot_mean = tot_max = 0.0
wg_total = (st[:, :, idx[-1]].max(axis=1) - st[:, :, 0].min(axis=1))
for k, nm in enumerate(names):
    d = st[:, :, idx[k + 1]] - st[:, :, idx[k]]
    print(f"  {nm:30s} mean over waves {d.mean() / 1e3:8.1f} k   slowest wave {d.max(axis=1).mean() / 1e3:8.1f} k   fastest {d.min(axis=1).mean() / 1e3:8.1f} k")
    tot_mean += d.mean()
print(f"  workgroup first-in -> last-out {wg_total.mean() / 1e3:.1f} k ticks (sum of phase means {tot_mean / 1e3:.1f} k); runs per wave: mean {st[:, :, 30].mean():.2f}, max over a workgroup's waves {st[:, :, 30].max(axis=1).mean():.2f}")
