"""Real-basis Wigner 3j tensors for (1,1,0), (1,1,1), (1,2,1) derived from first principles (TEST INFRASTRUCTURE, oracle/).

Why: oracle/thirdparty.py restates e3nn 0.5.1's `FullTensorProduct` / `FullyConnectedTensorProduct` / `spherical_harmonics`
from memory (e3nn is not vendored in the reference and not installable here).  This module replaces "recalled" by "derived
in-repo" for the algebra those restatements rest on:

  1. SU(2) Clebsch-Gordan coefficients <l1 m1 l2 m2 | l3 m3> in the Condon-Shortley convention, exact, from sympy
     (sympy.physics.quantum.cg.CG).
  2. The unitary change of basis real -> complex spherical basis, Q(l), in the construction e3nn documents for
     `o3.wigner_3j` (rows m = -l..l:  m<0: (|m>_c = (|-|m|>_r' ... ) written out in `real_to_complex`; overall phase (-i)^l so that
     the transformed coupling tensor is real).
  3. C[j,l,m] = sum Q1[i,j] Q2[k,l] conj(Q3)[m,n]^T C_su2[i,k,n], which must come out purely real (asserted), normalised to
     Frobenius norm 1 - the defining properties of `o3.wigner_3j(l1,l2,l3)`: an SO(3)-invariant tensor of unit norm in the
     real basis.  For (1,1,0), (1,1,1), (1,2,1) the invariant subspace is ONE-dimensional, so the tensor is determined up
     to a sign by invariance + normalisation alone; steps 1-3 fix the sign under e3nn's documented construction.
  4. The l=2 real basis is tied to vectors the way e3nn generates its harmonics: Y_2(v) is proportional, with a POSITIVE
     factor, to sum_{p,q} W(1,2,1)[p,j,q] v_p v_q (harmonics generated from the library's own 3j tensors).  With that, the
     sign of the `1o (x) 2e -> 1o` block of FullTensorProduct does not depend on any phase choice:
     out(a = v, Y_2(v)) . v = sqrt(3) c sum_j (sum_ik W[i,j,k] v_i v_k)^2 > 0.

Index convention: component i of an l=1 irrep is the vector component (x, y, z) in e3nn's labelling, which is the standard
real basis (m = -1, 0, +1) ~ (y, z, x)_std under the cyclic relabelling x_std = z, y_std = x, z_std = y (a proper rotation:
delta and epsilon are unchanged by it).  oracle/thirdparty._y2_norm is the standard real l=2 basis under the same
relabelling (checked in tests/test_thirdparty_algebra.py).

Residual risk after this derivation (named, not removable offline): that e3nn 0.5.1 (a) orders l=2 components m=-2..2 in
this basis, (b) uses 'component' normalisation factors sqrt(2l+1) as restated, and (c) numbers FullTensorProduct's output
blocks and FullyConnectedTensorProduct's weights as restated in oracle/thirdparty.py.
oracle/check_thirdparty.py runs the comparison against the real packages whenever they can be imported.
"""
import functools

import numpy as np


def su2_cg(l1, l2, l3):
    """C[l1+m1, l2+m2, l3+m3] = <l1 m1 l2 m2 | l3 m3>, Condon-Shortley, exact (sympy) -> float64."""
    from sympy import N
    from sympy.physics.quantum.cg import CG
    C = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            m3 = m1 + m2
            if abs(m3) <= l3:
                C[l1 + m1, l2 + m2, l3 + m3] = float(N(CG(l1, m1, l2, m2, l3, m3).doit(), 30))
    return C


def real_to_complex(l):
    """Q[l+m_complex, l+m_real]: complex spherical basis vector m in terms of the real basis, times (-i)^l.
    m<0:  Y_m = ( Y^r_{|m|} - i Y^r_{-|m|} ) / sqrt(2);  m=0: Y_0 = Y^r_0;  m>0: Y_m = (-1)^m ( Y^r_{m} + i Y^r_{-m} ) / sqrt(2)."""
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    s = 1.0 / np.sqrt(2.0)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = s
        q[l + m, l - abs(m)] = -1j * s
    q[l, l] = 1.0
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m * s
        q[l + m, l - abs(m)] = 1j * (-1) ** m * s
    return (-1j) ** l * q


@functools.lru_cache(maxsize=None)
def wigner_3j_real(l1, l2, l3):
    """Real-basis 3j tensor [2l1+1, 2l2+1, 2l3+1], Frobenius norm 1 (steps 1-3 of the module docstring)."""
    C = su2_cg(l1, l2, l3).astype(np.complex128)
    Q1, Q2, Q3 = real_to_complex(l1), real_to_complex(l2), real_to_complex(l3)
    R = np.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, np.conj(Q3.T), C)
    assert np.abs(R.imag).max() < 1e-12 or np.abs(R.real).max() < 1e-12, "coupling tensor is neither real nor imaginary"
    R = R.real if np.abs(R.imag).max() < 1e-12 else R.imag
    return R / np.linalg.norm(R)


def rotation_matrices(seed=0, n=4):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        out.append(np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                             [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                             [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]))
    return out


def y2_from_w3j(v):
    """l=2 harmonics generated from the 3j tensor itself: Y_2(v)_j = c * sum_pq W121[p,j,q] v_p v_q with c > 0 chosen so
    that |Y_2| = 1 for unit v ('norm' normalisation)."""
    W = wigner_3j_real(1, 2, 1)
    y = np.einsum("pjq,...p,...q->...j", W, v, v)
    return y / np.linalg.norm(y, axis=-1, keepdims=True)


def summary():
    W110, W111, W121 = wigner_3j_real(1, 1, 0), wigner_3j_real(1, 1, 1), wigner_3j_real(1, 2, 1)
    eps = np.zeros((3, 3, 3))
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[i, j, k], eps[i, k, j] = 1.0, -1.0
    return {"w110_vs_delta_over_sqrt3": float(np.abs(W110[:, :, 0] - np.eye(3) / np.sqrt(3)).max()),
            "w111_vs_eps_over_sqrt6 (up to sign)": float(min(np.abs(W111 - eps / np.sqrt(6)).max(), np.abs(W111 + eps / np.sqrt(6)).max())),
            "w121_symmetric_in_(i,k)": float(np.abs(W121 - W121.transpose(2, 1, 0)).max())}


if __name__ == "__main__":
    import json
    print(json.dumps(summary(), indent=1))
