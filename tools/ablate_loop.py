#!/usr/bin/env python3
"""Diagnostic: time the layer-3 conv launch of the DIRECT path (model.factorize_min_degree = 0: all nine convs in one
64-edge launch; all four blocks / scalar block only) with the product library and with the timing-only ablation builds
(1: no weight loads in the scalar main loop, 2: no LDS A reads).  One process per variant."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import copy, os, sys, torch
sys.path.insert(0, %r)
import bench
from diffdock_pocket_amd import score_model as sm
from diffdock_pocket_amd.diffusion import get_t_schedule
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
from diffdock_pocket_amd.synthetic import make_3dpf_complex
dev = torch.device("cuda:0")
model, kw = bench.build_model("cfg2", False, dev)
model.factorize_min_degree = 0      # every conv on the direct path: the launch this tool dissects
g = make_3dpf_complex(seed=0, flexible_sidechains=False)
smp = Sampler(model, g, 40, dev, SamplerConfig(flexible_sidechains=False), seed=0)
smp.randomize()
orig = sm._launch_convs
saved = {}
def hooked(spec, tasks, **kw):
    if (not spec.factorized) and len(spec.blocks) == 4 and len(tasks) == 9 and spec.blocks[3].ntiles == 140 and "t" not in saved:
        saved["t"] = (spec, list(tasks))
    orig(spec, tasks, **kw)
sm._launch_convs = hooked
smp.step(0, get_t_schedule(20))
torch.cuda.synchronize()
spec, tasks = saved["t"]
def timeit(blocks):
    sp = copy.copy(spec); sp.blocks = [spec.blocks[i] for i in blocks]
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(sp, tasks); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
wg = sum((t.n_edges + 63) // 64 for t in tasks)
for sel in ([0, 1, 2, 3], [0], [0, 3]):
    ms = timeit(sel)
    tiles = sum(spec.blocks[i].ntiles for i in sel)
    mfma = wg * (tiles * 2 * (spec.hp // 8) * 4 + spec.nct1 * 2 * (spec.kp1 // 8) * 4)
    print("  blocks %%-14s %%7.2f ms  executed MFMA %%6.1f TFLOP/s  (MFMA-pipe busy %%4.1f %%%%)" %% (sel, ms, mfma * 4096 / ms / 1e9, 100 * mfma * 64 / (ms * 1e-3 * 2.39e9 * 1024)))
'''

from diffdock_pocket_amd import build  # noqa: E402

for name, lib in (("product", build.build(verbose=False)), ("no weight loads", build.build(ablate=1)),
                  ("no LDS A reads", build.build(ablate=2))):
    print(name)
    env = dict(os.environ, DDP_HIP_LIB=lib)
    subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, check=True)
