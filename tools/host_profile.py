#!/usr/bin/env python3
"""Host-side profile of a denoising step at a host-bound size (5 samples: one rank's shard of configs[3]): cProfile over 10
steps, top functions by cumulative and by own time.  Usage on the GPU box:  python tools/host_profile.py [--samples 5]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

n = int(sys.argv[sys.argv.index("--samples") + 1]) if "--samples" in sys.argv else 5
dev = torch.device("cuda:0")
model, kw = bench.build_model("cfg2", False, dev)
g = make_3dpf_complex(seed=0, flexible_sidechains=False)
smp = Sampler(model, g, n, dev, SamplerConfig(flexible_sidechains=False), seed=0)
smp.randomize()
sched = get_t_schedule(20)
for i in range(4):
    smp.step(i, sched)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(4, 14):
    smp.step(i, sched)
torch.cuda.synchronize()
print(f"{n} samples: {(time.perf_counter() - t0) * 100:.2f} ms per step (wall, 10 steps)")
pr = cProfile.Profile()
pr.enable()
for i in range(4, 14):
    smp.step(i, sched)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(35)
