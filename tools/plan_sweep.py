#!/usr/bin/env python3
"""Diagnostic: ms per denoising step over the full 20-step schedule for several batch sizes and `plan_min_edges` thresholds
(0 = side-stream plans always on, 1e9 = never), plus the per-step times of one run (synchronised after every step: the
early schedule positions have the widest cross cutoffs and the most edges).  Usage on the GPU box:  python tools/plan_sweep.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, bench
from diffdock_pocket_amd.diffusion import get_t_schedule
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
from diffdock_pocket_amd.synthetic import make_3dpf_complex
dev = torch.device("cuda:0")
model, kw = bench.build_model("cfg2", False, dev)
g = make_3dpf_complex(seed=0, flexible_sidechains=False)
sched = get_t_schedule(20)
for n in (16, 40):
    for pm in (0, 100_000, 1_000_000_000):
        model.plan_min_edges = pm
        smp = Sampler(model, g, n, dev, SamplerConfig(flexible_sidechains=False), seed=0)
        smp.randomize()
        for i in range(2): smp.step((i * 10) % 20, sched)
        smp = Sampler(model, g, n, dev, SamplerConfig(flexible_sidechains=False), seed=0)
        smp.randomize()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        per = []
        for i in range(20):
            smp.step(i, sched)
            if pm == 100_000:
                torch.cuda.synchronize(); per.append(time.perf_counter())
        torch.cuda.synchronize()
        print(n, pm, round((time.perf_counter() - t0) / 20 * 1e3, 2), "ms/step (full schedule)", flush=True)
        if per:
            import numpy as np
            d = np.diff(np.array([t0] + per)) * 1e3
            print("   per step (synchronised):", " ".join(f"{x:.1f}" for x in d), flush=True)
