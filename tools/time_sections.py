"""Per-section GPU / host time of the denoising step (diagnostic; not part of the product path).

    python tools/time_sections.py [--samples 40] [--flex] [--iters 5]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.score_model import SectionTimer  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=40)
    ap.add_argument("--flex", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, _ = bench.build_model("cfg2", a.flex, dev)
    cg = make_3dpf_complex(seed=0, flexible_sidechains=a.flex)
    sampler = Sampler(model, cg, a.samples, dev, SamplerConfig(inference_steps=20, flexible_sidechains=a.flex), seed=0)
    sampler.randomize()
    sched = get_t_schedule(20)
    for i in range(2):
        sampler.step(i, sched)
    torch.cuda.synchronize()
    model.section_timer = st = SectionTimer()
    t0 = time.perf_counter()
    for i in range(a.iters):
        sampler.step(2 + i, sched)
    s = st.summary()
    wall = (time.perf_counter() - t0) / a.iters * 1e3
    tg = sum(v[0] for v in s.values()) / a.iters
    th = sum(v[1] for v in s.values()) / a.iters
    print(f"{'section':16s} {'gpu ms':>9s} {'host ms':>9s}")
    for k, (g, h) in s.items():
        print(f"{k:16s} {g / a.iters:9.3f} {h / a.iters:9.3f}")
    print(f"{'forward total':16s} {tg:9.3f} {th:9.3f}")
    print(f"step wall {wall:.3f} ms (forward + pose update)")


if __name__ == "__main__":
    main()
