"""Size-independent properties of the hot path at the BASELINE configuration (3dpf, 40 samples, cfg2: ns=60 nv=10 L=6,
flexible side chains): the oracle cannot run this size in test time, the symmetries of the score model can.

  * SE(3) equivariance (the reason the reference builds on e3nn irreps): moving the whole complex by a rotation R and a
    translation changes tr / rot scores into R tr / R rot (proper rotation: pseudo-vectors turn the same way) and leaves
    the torsion scores alone;
  * sample independence: a sample's scores do not depend on which other samples share its batch (what sharding over GPUs
    and the resident groups of the sampler rely on);
  * the exact-work eliminations (layer-0 sharing, dead-output pruning, factorisation) do not change the result at full size.
Tolerances are fp32 rounding amplified by a 6-layer network with random weights, relative to the largest component."""
import math

import pytest
import torch

import bench
from diffdock_pocket_amd.batch import collate, set_time
from diffdock_pocket_amd.synthetic import make_3dpf_complex
from helpers import rel_err

pytestmark = pytest.mark.gpu
N = 40


@pytest.fixture(scope="module")
def setup():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    dev = torch.device("cuda:0")
    model, _ = bench.build_model("cfg2", True, dev)
    g = make_3dpf_complex(seed=0, flexible_sidechains=True)
    gen = torch.Generator().manual_seed(5)
    graphs = []
    for _ in range(N):   # N poses of the complex: rigidly moved ligands (different edge sets per sample)
        c = g.clone()
        q = torch.randn(4, generator=gen)
        q = q / q.norm()
        w, x, y, z = q.tolist()
        R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        p = c["ligand"].pos
        c["ligand"].pos = (p - p.mean(0)) @ R.T + p.mean(0) + torch.randn(1, 3, generator=gen) * 1.5
        graphs.append(c)
    return dev, model, graphs


def _forward(model, graphs, dev, t=0.6, move=None):
    b = collate(graphs)
    set_time(b, t, t, t, t)
    if move is not None:
        R, tvec = move
        for k in ("ligand", "receptor", "atom"):
            b[k].pos = b[k].pos @ R.T + tvec
    out = model(b.to(dev))
    torch.cuda.synchronize()
    return [o.float().cpu() for o in out]


def test_se3_equivariance_at_full_size(setup):
    dev, model, graphs = setup
    tr, rot, tor, sc = _forward(model, graphs, dev)
    assert tr.shape == (N, 3) and tor.numel() > 0 and sc.numel() > 0
    ax = torch.tensor([0.3, -0.5, 0.8])
    ax = ax / ax.norm()
    ang = 1.1
    K = torch.tensor([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = torch.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)
    tr2, rot2, tor2, sc2 = _forward(model, graphs, dev, move=(R, torch.tensor([[3.0, -2.0, 1.5]])))
    tol = 2e-3
    assert rel_err(tr2, tr @ R.T) < tol, rel_err(tr2, tr @ R.T)
    assert rel_err(rot2, rot @ R.T) < tol, rel_err(rot2, rot @ R.T)
    assert rel_err(tor2, tor) < tol and rel_err(sc2, sc) < tol, (rel_err(tor2, tor), rel_err(sc2, sc))
    # the scores are far from rotation invariant themselves (the check above is not vacuous)
    assert rel_err(tr2, tr) > 0.1


def test_samples_are_independent_at_full_size(setup):
    dev, model, graphs = setup
    tr, rot, tor, sc = _forward(model, graphs, dev)
    T, S = tor.numel() // N, sc.numel() // N
    lo, hi = 10, 25
    tr_s, rot_s, tor_s, sc_s = _forward(model, graphs[lo:hi], dev)
    tol = 1e-3
    assert rel_err(tr_s, tr[lo:hi]) < tol and rel_err(rot_s, rot[lo:hi]) < tol
    assert rel_err(tor_s, tor[lo * T:hi * T]) < tol and rel_err(sc_s, sc[lo * S:hi * S]) < tol
    # reversed order of the graphs -> reversed outputs
    tr_r, rot_r, tor_r, sc_r = _forward(model, graphs[::-1], dev)
    assert rel_err(tr_r.flip(0), tr) < tol and rel_err(rot_r.flip(0), rot) < tol
    assert rel_err(tor_r.reshape(N, T).flip(0).reshape(-1), tor) < tol


def test_work_eliminations_are_exact_at_full_size(setup):
    dev, model, graphs = setup
    want = _forward(model, graphs, dev)
    for a, b in zip(want, _forward(model, graphs, dev)):       # repeated call: bitwise (no float atomics on the path)
        assert torch.equal(a, b)
    saved = (model.share_layer0, model.prune_last_receptor_layer, model.factorize_min_degree)
    try:
        model.exact_sizes = True                # every device-side list size read back, exact grids: same bits
        for a, b in zip(want, _forward(model, graphs, dev)):
            assert torch.equal(a, b)
        model.exact_sizes = False
        model.share_layer0, model.prune_last_receptor_layer = False, False
        plain = _forward(model, graphs, dev)
        model.factorize_min_degree = 0          # every conv on the direct per-edge MFMA path
        direct = _forward(model, graphs, dev)
    finally:
        model.exact_sizes = False
        model.share_layer0, model.prune_last_receptor_layer, model.factorize_min_degree = saved
    for a, b, c in zip(want, plain, direct):
        assert rel_err(a, b) < 1e-4, rel_err(a, b)
        assert rel_err(a, c) < 1e-3, rel_err(a, c)
