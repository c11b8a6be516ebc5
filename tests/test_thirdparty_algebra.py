"""CPU: the e3nn algebra restated in oracle/thirdparty.py (FullTensorProduct's 1o block, FullyConnectedTensorProduct's
1 x 1 -> 0 paths, the l<=2 harmonics) against real-basis Wigner 3j tensors DERIVED in-repo from Condon-Shortley
Clebsch-Gordan coefficients (oracle/derive_w3j.py, sympy) - "derived", no longer "recalled"; residual risk in that
module's docstring."""
import math

import numpy as np
import torch

from oracle import derive_w3j as D
from oracle import thirdparty as tp


def _unit(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True), rng


def test_3j_tensors_are_invariant_unit_norm_and_have_the_closed_forms():
    W110, W111, W121 = D.wigner_3j_real(1, 1, 0), D.wigner_3j_real(1, 1, 1), D.wigner_3j_real(1, 2, 1)
    for W in (W110, W111, W121):
        assert abs(np.linalg.norm(W) - 1.0) < 1e-12
    assert np.abs(W110[:, :, 0] - np.eye(3) / math.sqrt(3)).max() < 1e-12          # (1,1,0) = +delta / sqrt(3)
    eps = np.zeros((3, 3, 3))
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[i, j, k], eps[i, k, j] = 1.0, -1.0
    assert np.abs(np.abs(W111) - np.abs(eps) / math.sqrt(6)).max() < 1e-12         # (1,1,1) = +-eps / sqrt(6)
    assert np.abs(W121 - W121.transpose(2, 1, 0)).max() < 1e-12                    # l1+l2+l3 even: symmetric in the two l=1 slots
    # SO(3) invariance with the l=2 representation induced by the harmonics generated from W121 itself
    v, rng = _unit(6, 0)
    for R in D.rotation_matrices(seed=1, n=3):
        a, b = rng.normal(size=3), rng.normal(size=3)
        assert np.abs(np.einsum("ijk,i,j->k", W111, R @ a, R @ b) - R @ np.einsum("ijk,i,j->k", W111, a, b)).max() < 1e-12
        y, yr = D.y2_from_w3j(v), D.y2_from_w3j(v @ R.T)
        t = np.einsum("ijk,i,nj->nk", W121, a, y)
        tr = np.einsum("ijk,i,nj->nk", W121, R @ a, yr)
        assert np.abs(tr - t @ R.T).max() < 1e-12


def test_restated_harmonics_are_the_ones_generated_by_the_3j_tensor():
    """Y_2(v) ~ +c sum W121[p,j,q] v_p v_q: same five components, same order, POSITIVE factor; 'component' normalisation
    = sqrt(2l+1) x unit-norm harmonics."""
    v, _ = _unit(9, 2)
    y_w = D.y2_from_w3j(v)
    y_tp = tp._y2_norm(torch.tensor(v)).numpy()
    assert np.abs(y_w - y_tp).max() < 1e-12
    sh = tp.spherical_harmonics("1x0e+1x1o", torch.tensor(v)).numpy()
    assert np.abs(sh[:, 0] - 1).max() < 1e-12 and np.abs(sh[:, 1:] - math.sqrt(3) * v).max() < 1e-12
    y2c = tp.spherical_harmonics("2e", torch.tensor(v)).numpy()
    assert np.abs(np.linalg.norm(y2c, axis=1) - math.sqrt(5)).max() < 1e-12


def test_full_tensor_product_1o_block_equals_the_derived_coupling():
    """1o block of FullTensorProduct(sh, Y2) = sqrt(2*1+1) * sum_ij W121[i,j,k] a_i b_j ('component' normalisation of the
    output irrep), sign and magnitude (KAPPA = 3/sqrt(10)); and its sign is phase independent: out(a=v) . v > 0."""
    v, rng = _unit(7, 3)
    a = rng.normal(size=(7, 3))
    sh = torch.tensor(np.concatenate([np.ones((7, 1)), a], 1))
    y2c = tp.spherical_harmonics("2e", torch.tensor(v))
    got = tp.FullTensorProduct("1x0e+1x1o", "2e")(sh, y2c)[:, :3].numpy()
    want = math.sqrt(3) * np.einsum("ijk,ni,nj->nk", D.wigner_3j_real(1, 2, 1), a, y2c.numpy())
    assert np.abs(got - want).max() < 1e-12
    shv = torch.tensor(np.concatenate([np.ones((7, 1)), v], 1))
    assert (np.einsum("nk,nk->n", tp.FullTensorProduct("1x0e+1x1o", "2e")(shv, y2c)[:, :3].numpy(), v) > 0).all()


def test_fctp_scalar_paths_equal_the_derived_coupling():
    """The torsion heads' FullyConnectedTensorProduct keeps two 1 x 1 -> 0 paths; each = path_weight * sum_uw w[u,w] *
    sum_ij W110[i,j,0] a_u,i b_j with W110 = delta / sqrt(3) and path weight sqrt(1 / mul_in) (one path per output)."""
    rng = np.random.default_rng(4)
    m1o, m1e, ns = 3, 2, 4
    tpm = tp.FullyConnectedTensorProduct(f"5x0e+{m1o}x1o+{m1e}x1e+5x0o", "1x1o+1x2e+1x2o+1x3o", f"{ns}x0o+{ns}x0e")
    assert tpm.weight_numel == (m1o + m1e) * ns
    x = torch.tensor(rng.normal(size=(6, 5 + 3 * m1o + 3 * m1e + 5)))
    y = torch.tensor(rng.normal(size=(6, 20)))
    w = torch.tensor(rng.normal(size=(6, tpm.weight_numel)))
    got = tpm(x, y, w).numpy()
    W110 = D.wigner_3j_real(1, 1, 0)[:, :, 0]
    a1o = x[:, 5:5 + 3 * m1o].reshape(6, m1o, 3).numpy()
    a1e = x[:, 5 + 3 * m1o:5 + 3 * m1o + 3 * m1e].reshape(6, m1e, 3).numpy()
    b = y[:, :3].numpy()
    w = w.numpy()
    # instruction order: (1o x 1o -> 0e) first, then (1e x 1o -> 0o); outputs laid out [0o | 0e]
    out0e = np.einsum("nui,ij,nj,nuw->nw", a1o, W110, b, w[:, :m1o * ns].reshape(6, m1o, ns)) / math.sqrt(m1o)
    out0o = np.einsum("nui,ij,nj,nuw->nw", a1e, W110, b, w[:, m1o * ns:].reshape(6, m1e, ns)) / math.sqrt(m1e)
    assert np.abs(got[:, ns:] - out0e).max() < 1e-12 and np.abs(got[:, :ns] - out0o).max() < 1e-12
