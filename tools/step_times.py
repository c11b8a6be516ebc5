"""ms per denoising step, launch by launch against hipGraph replay, for bench.py's jobs (python tools/step_times.py)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402

dev = torch.device("cuda:0")
sched = get_t_schedule(20)
for cfg, flex, n in (("cfg2", False, 40), ("cfg2", True, 40), ("cfg2", False, 5), ("cfg1", True, 4)):
    model, kw = bench.build_model(cfg, flex, dev)
    g = make_3dpf_complex(seed=0, flexible_sidechains=flex)
    res = {}
    for graph in (False, True):
        smp = Sampler(model, g, 40 if n == 5 else n, dev, SamplerConfig(inference_steps=20, flexible_sidechains=flex, hip_graph=graph), seed=0,
                      sample_slice=slice(0, n))
        smp.randomize()
        snap = smp.snapshot()
        for i in range(3):
            smp.step((i * 10) % 20, sched)
        smp.restore(snap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            smp.step(i, sched)
        torch.cuda.synchronize()
        res[graph] = ((time.perf_counter() - t0) / 20 * 1e3, bool(smp._graph), smp.lig_pos.clone())
    same = torch.equal(res[False][2], res[True][2])
    print(f"{cfg} flex={flex} n={n}: eager {res[False][0]:.2f} ms/step, graph {res[True][0]:.2f} ms/step (captured={res[True][1]}), "
          f"same poses bitwise: {same}", flush=True)
