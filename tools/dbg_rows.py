#!/usr/bin/env python3
"""Diagnostic: messages of one factorised conv through ddp_conv_rows vs the 32-edge kernel (launch.CONV_ROWS = False), by output
column block and by 32-edge row tile of the source-ordered list.  python tools/dbg_rows.py [layer E N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from diffdock_pocket_amd import graph as G  # noqa: E402
from diffdock_pocket_amd import launch as K  # noqa: E402
from diffdock_pocket_amd import packing as P  # noqa: E402
from diffdock_pocket_amd.score_model import TensorProductConvLayer  # noqa: E402

layer, E, N = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (3, 200, 23)))
ns, nv = 60, 10
torch.manual_seed(ns + layer + E)
mi, mo = P.irreps_muls(ns, nv, layer), P.irreps_muls(ns, nv, layer + 1)
spec = P.faster_tp_spec(mi, mo, 3 * ns)
blocks = [(m, d, s) for m, d, s in ((mo[0], 1, True), (mo[1], 3, False), (mo[2], 3, False), (mo[3], 1, False)) if m]
conv = TensorProductConvLayer(spec, blocks, spec_g=P.faster_tp_spec(mi, mo, 3 * ns, factorized=True))
dev = torch.device("cuda:0")
conv = conv.to(dev)
x = torch.randn(N, P.irreps_dim(mi)).to(dev)
ei = torch.stack([torch.randint(0, max(N - 1, 1), (E,)), torch.randint(0, N, (E,))]).to(dev)
ea = torch.randn(E, 3 * ns).to(dev)
v = torch.randn(E, 3)
v = v / v.norm(dim=1, keepdim=True)
sh = torch.cat([torch.ones(E, 1), 3 ** 0.5 * v], 1).to(dev)
csr = G.build_csr(ei[0].long(), ei[1].long(), N)
so = G.source_order(csr, N)
pk = conv.packed_g(dev)
out = {}
for rows in (False, True):
    K.CONV_ROWS = rows
    rk = K.rows_mode(pk)
    assert rk == rows
    g = conv.node_tensors(pk, x, rows=rk)
    msg = torch.full((csr.n_edges, spec.d_out), float("nan"), device=dev)
    if rk:
        w3 = ea.shape[1] // 3
        segs = [(ea[:, i * w3:], so.eid, ea.shape[1], w3) for i in range(3)]
    else:
        segs = [(ea, so.eid, ea.shape[1], ea.shape[1])]
    task = K.make_task(pk, x, x.shape[1], so, sh, segs, msg, g=g, rows=rk)
    K.launch_convs(conv.spec_g, [task], flops_spec=spec)
    torch.cuda.synchronize()
    out[rows] = msg.cpu()
a, b = out[False], out[True]
print("nan in rows result:", int(torch.isnan(b).sum()), "of", b.numel(), "| scale", float(a.abs().max()))
pos = so.pos.cpu().long() if so.pos is not None else torch.arange(E)
cols = [0]
for m, d, _ in blocks:
    cols.append(cols[-1] + m * d)
d = (a - b).abs()
d[torch.isnan(d)] = 1e9
for i in range(len(cols) - 1):
    blk = d[:, cols[i]:cols[i + 1]]
    print(f"columns [{cols[i]}, {cols[i + 1]}): max |diff| {float(blk.max()):.3e}")
for t0 in range(0, E, 32):
    rowsel = pos[t0:t0 + 32]
    print(f"row tile {t0 // 32}: " + " ".join(f"{float(d[rowsel][:, cols[i]:cols[i + 1]].max()):.1e}" for i in range(len(cols) - 1)))
src_sorted = so.src.cpu()[:E]
print("runs per 32-edge tile:", [int((src_sorted[t0:t0 + 32][1:] != src_sorted[t0:t0 + 32][:-1]).sum()) + 1 for t0 in range(0, E, 32)])
