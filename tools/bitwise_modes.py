"""Diagnostic: bitwise comparison of the forward at the BASELINE size between the modes of the dead-output walk (side stream /
in the front / off) and between repeated calls.  Everything must print zeros (it did not while the ligand centre was an
index_add_ of floats: atomics order).  python tools/bitwise_modes.py [--noflex]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench
from diffdock_pocket_amd.batch import collate, set_time
from diffdock_pocket_amd.synthetic import make_3dpf_complex
dev = torch.device("cuda:0")
flex = "--noflex" not in sys.argv
model, _ = bench.build_model("cfg2", flex, dev)
g = make_3dpf_complex(seed=0, flexible_sidechains=flex)
gen = torch.Generator().manual_seed(5)
graphs = []
for _ in range(40):
    c = g.clone()
    c["ligand"].pos = c["ligand"].pos + torch.randn(1, 3, generator=gen) * 1.5
    graphs.append(c)
def fwd():
    b = collate(graphs); set_time(b, 0.6, 0.6, 0.6, 0.6)
    out = model(b.to(dev)); torch.cuda.synchronize()
    return [o.float().cpu() for o in out]
def cmp(a, b, tag):
    print(tag, [float((x - y).abs().max()) if x.numel() else 0.0 for x, y in zip(a, b)])
a1 = fwd(); a2 = fwd(); cmp(a1, a2, "async vs async")
model.prune_async = False
s1 = fwd(); s2 = fwd(); cmp(s1, s2, "sync vs sync"); cmp(a1, s1, "async vs sync")
model.prune_last_receptor_layer = False
n1 = fwd(); cmp(s1, n1, "sync-pruned vs unpruned"); cmp(a1, n1, "async-pruned vs unpruned")
