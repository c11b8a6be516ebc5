"""Per-launch conv times of the forward on a FIXED batch (poses not updated): for timing-only kernel variants whose results
are wrong by construction (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from diffdock_pocket_amd import score_model as sm  # noqa: E402
from diffdock_pocket_amd.batch import set_time  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

dev = torch.device("cuda:0")
model, _ = bench.build_model("cfg2", False, dev)
cg = make_3dpf_complex(seed=0, flexible_sidechains=False)
smp = Sampler(model, cg, 40, dev, SamplerConfig(inference_steps=20, flexible_sidechains=False), seed=0)
smp.randomize()
smp.lig_pos = smp.lig_pos * 0 + smp.batch["ligand"].pos.reshape(40, -1, 3) + torch.randn(40, 1, 3, device=dev) * 1.0   # near the pocket
b = smp.batch
b["ligand"].pos = smp.lig_pos.reshape(-1, 3)
set_time(b, 0.5, 0.5, 0.5, 0.5, device=dev)
for _ in range(2):
    model(b)
prof = sm.ConvProfiler()
sm.set_conv_profiler(prof)
for _ in range(3):
    model(b)
torch.cuda.synchronize()
sm.set_conv_profiler(None)
tot = {}
for (e0, e1), k in zip(prof.events, prof.kernel):
    tot[k] = tot.get(k, 0.0) + e0.elapsed_time(e1) / 3
print({k: round(v, 3) for k, v in tot.items()}, "edges", model.last_stats["E_lr"], model.last_stats["E_la"])
