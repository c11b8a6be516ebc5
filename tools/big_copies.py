#!/usr/bin/env python3
"""Diagnostic: which torch ops issue the large device-to-device copies of a step (rocclr copyBuffer launches > 10 us)?
Prints op, input shapes and the chain of enclosing ops.  Usage on the GPU box:  python tools/big_copies.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

dev = torch.device("cuda:0")
model, kw = bench.build_model("cfg2", False, dev)
g = make_3dpf_complex(seed=0, flexible_sidechains=False)
smp = Sampler(model, g, 5, dev, SamplerConfig(flexible_sidechains=False, hip_graph=False), seed=0)
smp.randomize()
sched = get_t_schedule(20)
for i in range(3):
    smp.step(i, sched)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    smp.step(3, sched)
    torch.cuda.synchronize()
rows = []
for f in prof.profiler.function_events:
    if f.device_type != torch.autograd.DeviceType.CPU or not f.kernels:
        continue
    if any(c.kernels for c in f.cpu_children):
        continue
    for k in f.kernels:
        if ("copyBuffer" in k.name or "Memcpy" in k.name or "fillBuffer" in k.name) and k.duration > 0:
            chain, p = [], f.cpu_parent
            while p is not None and len(chain) < 4:
                chain.append(p.name)
                p = p.cpu_parent
            rows.append((f.time_range.start, k.duration, k.name[:28], f.name, str(f.input_shapes)[:80], " < ".join(chain)))
rows.sort()
import collections  # noqa: E402
agg = collections.Counter()
dur = collections.Counter()
for r in rows:
    agg[(r[2], r[3], r[4], r[5])] += 1
    dur[(r[2], r[3], r[4], r[5])] += r[1]
for k, c in agg.most_common(40):
    print(f"{c:4d} x {dur[k] / c:6.1f} us  {k[0]:26s} {k[1]:20s} {k[2]}  | {k[3]}")
print("copies / fills in the step:", len(rows), "total us:", sum(r[1] for r in rows))
# every device activity of the step that is a runtime copy, whatever the host op (ctypes launches have no aten parent)
n_all = sum(1 for e in prof.profiler.kineto_results.events() if "copyBuffer" in e.name() or "Memcpy" in e.name())
print("runtime copies seen by kineto:", n_all)
