import os, sys, time, math
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffdock_pocket_amd import sampler as S
from diffdock_pocket_amd.synthetic import make_3dpf_complex
dev = torch.device("cuda:0")
g = make_3dpf_complex(seed=0, flexible_sidechains=False)
n, nl = 40, g["ligand"].pos.shape[0]
em = g["ligand"].edge_mask.bool()
bonds = g["ligand", "ligand"].edge_index.t()[em].clone()
import numpy as np
mr = g["ligand"].mask_rotate
mask = torch.as_tensor(np.asarray(mr if isinstance(mr, np.ndarray) else mr[0])).bool().to(dev)
T = bonds.shape[0]
pos = g["ligand"].pos.unsqueeze(0).repeat(n, 1, 1).to(dev) + torch.randn(n, 1, 3, device=dev)
tr, rot, tor = torch.randn(n, 3, device=dev) * .3, torch.randn(n, 3, device=dev) * .2, torch.randn(n, T, device=dev) * .3
def timeit(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print("n_lig", nl, "T", T)
print("modify_conformer total", timeit(lambda: S.modify_conformer(pos, tr, rot, tor, bonds, mask)))
rigid = S.modify_conformer(pos, tr, rot, None, bonds, mask)
print("rigid part", timeit(lambda: S.modify_conformer(pos, tr, rot, None, bonds, mask)))
print("apply_torsions", timeit(lambda: S.apply_torsions(rigid, bonds, mask, tor)))
flex = S.apply_torsions(rigid, bonds, mask, tor)
print("kabsch", timeit(lambda: S.kabsch(flex, rigid)))
gen = torch.Generator().manual_seed(0)
print("cpu randn + H2D x3", timeit(lambda: [torch.randn((n, 3), generator=gen).to(dev), torch.randn((n, 3), generator=gen).to(dev), torch.randn((n, T), generator=gen).to(dev)]))
