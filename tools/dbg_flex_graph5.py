import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from diffdock_pocket_amd.diffusion import get_t_schedule
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
from diffdock_pocket_amd.synthetic import make_3dpf_complex
dev = torch.device("cuda:0")
sched = get_t_schedule(20)
class Stop(Exception):
    pass
class Marks:
    def __init__(self, stop):
        self.stop = stop
    def mark(self, name):
        if name == self.stop:
            raise Stop()
def run(stop):
    model, kw = bench.build_model("cfg2", True, dev)
    g = make_3dpf_complex(seed=0, flexible_sidechains=True)
    smp = Sampler(model, g, 8, dev, SamplerConfig(inference_steps=20, flexible_sidechains=True), seed=0)
    smp.randomize()
    for i in (0, 1, 2):
        smp.step(i, sched)
    if stop is not None:
        model.section_timer = Marks(stop)
        try:
            smp.scores(float(sched[10]))
        except Stop:
            pass
        model.section_timer = None
    smp.step(10, sched)
    torch.cuda.synchronize()
    return smp.lig_pos.clone()
base = run(None)
for stop in ("start", "node_embed", "searches", "edge_featurize", "views", "lists", "conv_prep", "tor_heads", "never"):
    got = run(stop)
    print(f"eager forward up to {stop:15s}: max diff {float((got - base).abs().max()):.3e}", flush=True)
