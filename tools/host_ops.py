#!/usr/bin/env python3
"""Which lines of the host glue launch the non-ddp device work of a denoising step?

Runs three steps of the bench workload under torch.profiler (with_stack) and attributes every device kernel / memcpy /
memset that is not a ddp_* kernel to the innermost frame inside diffdock_pocket_amd/ that caused it.  Prints launches and
device microseconds per step, per source line.  Usage on the GPU box:  python tools/host_ops.py [--flex] [--top 60]"""
import collections
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

flex = "--flex" in sys.argv
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
dev = torch.device("cuda:0")
model, kw = bench.build_model("cfg2", flex, dev)
g = make_3dpf_complex(seed=0, flexible_sidechains=flex)
smp = Sampler(model, g, 40, dev, SamplerConfig(flexible_sidechains=flex), seed=0)
smp.randomize()
sched = get_t_schedule(20)
for i in range(3):
    smp.step(i, sched)
torch.cuda.synchronize()
STEPS = 3
import traceback  # noqa: E402

from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402


class Count(TorchDispatchMode):
    """Every aten call that reaches the dispatcher, keyed by the innermost frame inside the package."""

    def __init__(self):
        super().__init__()
        self.rows = collections.defaultdict(collections.Counter)
        self.events = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        where = "(outside the package)"
        for fr in reversed(traceback.extract_stack(limit=14)):
            if "diffdock_pocket_amd/" in fr.filename:
                where = f"{fr.filename.split('diffdock_pocket_amd/')[1]}:{fr.lineno} {fr.line[:70] if fr.line else ''}"
                break
        self.rows[where][func.__name__] += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = func(*args, **(kwargs or {}))
        e1.record()
        self.events.append((where, e0, e1))
        return out


SKIP = ("view", "reshape", "expand", "as_strided", "slice", "select", "unsqueeze", "squeeze", "transpose", "t.default", "permute",
        "detach", "alias", "_unsafe_view", "unbind", "split", "narrow", "empty", "size", "stride", "is_", "sym_", "dim", "numel",
        "_local_scalar_dense", "item", "lift_fresh", "record_stream", "unfold", "chunk", "contiguous")
with Count() as cnt:
    for i in range(3, 3 + STEPS):
        smp.step(i, sched)
    torch.cuda.synchronize()
us = collections.Counter()
for where, e0, e1 in cnt.events:   # stream time between the op's first and last kernel (on the stream the op ran on)
    us[where] += e0.elapsed_time(e1) * 1e3
tot = 0
out = []
for where, c in cnt.rows.items():
    n = sum(v for k, v in c.items() if not any(k.startswith(s_) or s_ in k.split(".")[0] for s_ in SKIP))
    if n:
        out.append((n, where, c))
        tot += n
print(f"aten calls per step that launch device work (views / metadata ops excluded): {tot / STEPS:.1f}")
for n, where, c in sorted(out, key=lambda r: -r[0])[:top]:
    ops = ", ".join(f"{k.split('.')[0]}x{v // STEPS}" for k, v in c.most_common(5) if not any(k.startswith(s_) for s_ in SKIP))
    print(f"{n / STEPS:7.1f} {us[where] / STEPS:8.1f}us  {where}   [{ops}]")
print("\nby stream time:")
for where, u in us.most_common(40):
    print(f"{u / STEPS:8.1f}us  {where}")
sys.exit(0)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(3, 3 + STEPS):
        smp.step(i, sched)
    torch.cuda.synchronize()

ev = prof.profiler.kineto_results.events()
# device activities, by correlation id
dev_by_corr = collections.defaultdict(list)
for e in ev:
    if e.device_type() == torch.autograd.DeviceType.CUDA:
        dev_by_corr[e.linked_correlation_id() or e.correlation_id()].append(e)
rows = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
unattributed = [0, 0.0]
fe = prof.profiler.function_events
for f in fe:
    if f.device_type != torch.autograd.DeviceType.CPU or not f.kernels:
        continue
    ks = [k for k in f.kernels if not k.name.startswith("ddp_") and "ddp_" not in k.name]
    if not ks:
        continue
    # only leaf ops (an op whose child also lists the kernels would double count): take ops without cpu_children that have kernels
    if any(c.kernels for c in f.cpu_children):
        continue
    where = None
    for fr in (f.stack or []):
        if "diffdock_pocket_amd/" in fr:
            where = fr.split("diffdock_pocket_amd/")[1].strip()
            break
    us = sum(k.duration for k in ks)
    if where is None:
        unattributed[0] += len(ks)
        unattributed[1] += us
        where = "(outside the package) " + f.name
    r = rows[where]
    r[0] += len(ks)
    r[1] += us
    r[2][f.name] += len(ks)
tot_n = sum(r[0] for r in rows.values())
tot_us = sum(r[1] for r in rows.values())
print(f"non-ddp device activities per step: {tot_n / STEPS:.1f} launches, {tot_us / STEPS / 1e3:.3f} ms device time")
print(f"{'launches':>9s} {'us':>9s}  line (ops)")
for where, r in sorted(rows.items(), key=lambda kv: -kv[1][0])[:top]:
    ops = ", ".join(f"{n}x{c // STEPS if c >= STEPS else c}" for n, c in r[2].most_common(4))
    print(f"{r[0] / STEPS:9.1f} {r[1] / STEPS:9.1f}  {where}  [{ops}]")
