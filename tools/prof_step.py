"""torch.profiler view of one denoising step (diagnostic): which host ops launch the device copies / tiny kernels."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

dev = torch.device("cuda:0")
model, _ = bench.build_model("cfg2", False, dev)
cg = make_3dpf_complex(seed=0, flexible_sidechains=False)
smp = Sampler(model, cg, 40, dev, SamplerConfig(inference_steps=20, flexible_sidechains=False), seed=0)
smp.randomize()
sched = get_t_schedule(20)
for i in range(3):
    smp.step(i, sched)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    smp.step(3, sched)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
ev = [e for e in prof.events() if "emcpy" in e.name or "copy_" in e.name or "item" in e.name or "nonzero" in e.name]
from collections import Counter
print(Counter(e.name for e in ev).most_common(20))

print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=0, max_name_column_width=30)[:200])
rows = []
for e in prof.key_averages(group_by_stack_n=8):
    if e.key in ("aten::copy_", "aten::item", "aten::_local_scalar_dense", "aten::nonzero", "aten::to", "aten::_to_copy", "aten::contiguous", "aten::clone"):
        rows.append((e.count, e.key, [s_ for s_ in e.stack if "diffdock_pocket_amd" in s_ or "bench" in s_][:3]))
rows.sort(key=lambda r: -r[0])
for r in rows[:40]:
    print(r)
