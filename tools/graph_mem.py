"""Diagnostic: device memory held after a captured 40-sample step is dropped (hipGraph private pools)."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return f"used {(total - free) / 1e9:6.1f} GB  torch reserved {torch.cuda.memory_reserved() / 1e9:6.1f}  allocated {torch.cuda.memory_allocated() / 1e9:6.1f}"


def main():
    dev = torch.device("cuda:0")
    mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
    model, kw = bench.build_model("cfg2", False, dev)
    g = make_3dpf_complex(seed=0, flexible_sidechains=False)
    sched = get_t_schedule(20)
    print(mode, "start", used(), flush=True)
    for rep in range(4):
        smp = Sampler(model, g, 40, dev, SamplerConfig(inference_steps=20, flexible_sidechains=False, hip_graph=(mode != "nograph")), seed=0)
        smp.randomize()
        for i in range(5):
            smp.step(i, sched)
        print(mode, rep, "after 5 steps   ", used(), "graph", bool(smp._graph), flush=True)
        if mode == "reset" and smp._graph:
            smp._graph.reset()
        if mode == "close":
            smp.close()
        del smp
        model._static_cache = {}
        gc.collect()
        torch.cuda.empty_cache()
        print(mode, rep, "after del + empty", used(), flush=True)


if __name__ == "__main__":
    main()
