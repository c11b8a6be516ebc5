#!/usr/bin/env python3
"""Experiment: the 40-sample job as W groups of samples, each with its OWN captured step, replayed side by side on W HIP streams -
does one group's HBM-bound stage A overlap the other's matrix-bound conv launches?   python tools/two_graphs.py [ways] [flex]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

ways = int(sys.argv[1]) if len(sys.argv) > 1 else 2
flex = len(sys.argv) > 2 and sys.argv[2] == "flex"
dev = torch.device("cuda:0")
model, kw = bench.build_model("cfg2", flex, dev)
g = make_3dpf_complex(seed=0, flexible_sidechains=flex)
N = 40
sched = get_t_schedule(20)


def run(groups):
    cuts = [N * w // groups for w in range(groups + 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(groups)]
    parts = []
    for w in range(groups):
        model.cache_slot = w
        with torch.cuda.stream(streams[w]):
            s = Sampler(model, g, N, dev, SamplerConfig(flexible_sidechains=flex, hip_graph=True), seed=0, sample_slice=slice(cuts[w], cuts[w + 1]))
            s.randomize()
            for i in range(4):        # (the fourth step is replayed)
                s.step(i, sched)
        torch.cuda.synchronize()
        parts.append(s)
    times = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4, 20):
            for w in range(groups):
                model.cache_slot = w
                with torch.cuda.stream(streams[w]):
                    parts[w].step(i % 20, sched)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 16 * 1e3)
    model.cache_slot = 0
    return min(times), parts


for gcount in (1, ways, 1):
    ms, parts = run(gcount)
    print(f"{gcount} group(s) of {N // gcount} samples, one captured step each, side by side: {ms:.2f} ms per {N}-sample step", flush=True)
    del parts
    torch.cuda.synchronize()
