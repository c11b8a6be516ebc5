"""Diagnostic: mean launch time of the conv / stage-A kernels over one 20-step job with device-side counts + capacity grids
against host-known sizes + exact grids (model.exact_sizes).  python tools/exact_vs_capacity.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffdock_pocket_amd import score_model as sm  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

dev = torch.device("cuda:0")
sched = get_t_schedule(20)
model, kw = bench.build_model("cfg2", False, dev)
g = make_3dpf_complex(seed=0, flexible_sidechains=False)
for exact in (False, True, False, True):
    model.exact_sizes = exact
    smp = Sampler(model, g, 40, dev, SamplerConfig(inference_steps=20, flexible_sidechains=False, hip_graph=False), seed=0)
    smp.randomize()
    for i in range(2):
        smp.step(i, sched)
    snap = smp.snapshot() if False else None
    prof = sm.ConvProfiler()
    prof.hbm_on = True
    sm.set_conv_profiler(prof)
    for i in range(20):
        smp.step(i, sched)
    torch.cuda.synchronize()
    sm.set_conv_profiler(None)
    out = {}
    for k in ("ddp_conv32_kernel", "ddp_conv_messages_kernel"):
        n, _, ms = prof.summary(k)
        out[k] = round(ms / n, 4)
    n, by, ms = prof.hbm_summary("ddp_stage_a_mfma_kernel")
    out["stage_a ms/step"] = round(ms / 20, 3)
    n, by, ms = prof.hbm_summary("ddp_segment_reduce_kernel")
    out["reduce ms/step"] = round(ms / 20, 3)
    print("exact_sizes", exact, out, flush=True)
