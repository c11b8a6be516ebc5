"""Per-launch times of the conv kernels over one denoising step (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from diffdock_pocket_amd import score_model as sm  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

flex = "--flex" in sys.argv
dev = torch.device("cuda:0")
model, _ = bench.build_model("cfg2", flex, dev)
cg = make_3dpf_complex(seed=0, flexible_sidechains=flex)
smp = Sampler(model, cg, 40, dev, SamplerConfig(inference_steps=20, flexible_sidechains=flex), seed=0)
smp.randomize()
sched = get_t_schedule(20)
for i in range(3):
    smp.step(i, sched)
prof = sm.ConvProfiler()
sm.set_conv_profiler(prof)
smp.step(3, sched)
torch.cuda.synchronize()
sm.set_conv_profiler(None)
tot = 0.0
for (e0, e1), fl, ex, k in zip(prof.events, prof.flops, prof.executed, prof.kernel):
    ms = e0.elapsed_time(e1)
    tot += ms
    print(f"{k[4:12]}  {ms:7.3f} ms  algorithmic {fl / 1e9:8.1f} GFLOP ({fl / ms / 1e9:6.1f} TF/s)  executed {ex / ms / 1e9:6.1f} TF/s")
print(f"total {tot:.2f} ms")
