"""CPU restatement of the third-party ops the reference's score-model path calls.

TEST INFRASTRUCTURE (oracle/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package; the product (diffdock_pocket_amd/) never does.

The reference path (models/all_atom_score_model.py, models/score_model.py) reaches into packages that
are NOT vendored in /root/reference and NOT installed here:
    e3nn==0.5.1, torch-scatter==2.1.0, torch-cluster==1.6.1   (environment.yml:11-27)
Their published semantics are restated below from memory ("[recalled]" in SURVEY.md Appendix B).
PARITY UNPINNED for exactly these functions: the reference has no tests/golden vectors at these
boundaries and the real packages cannot be run here.  Everything written in the reference's own
files IS pinned, by golden vectors generated from those files under oracle/shim.py.

Call sites restated (reference file:line):
  spherical_harmonics   all_atom_score_model.py:394,418,481,508,534,556,570,579,598,613,633
  FullTensorProduct     all_atom_score_model.py:193,219,395,419
  FullyConnectedTP      score_model.py:98   (tor_bond_conv / sc_tor_bond_conv only, faster=False)
  BatchNorm             score_model.py:106,124
  scatter / scatter_mean score_model.py:117 ; all_atom_score_model.py:331,339
  radius / radius_graph / knn_graph   all_atom_score_model.py:457,524,545-564,607,627
"""
import math
import re

import torch

# --------------------------------------------------------------------------- irreps bookkeeping


class Irrep(tuple):
    """(l, p) with p = +1 (even) / -1 (odd)."""

    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            m = re.fullmatch(r"\s*(\d+)([eo])\s*", l)
            l, p = int(m.group(1)), (1 if m.group(2) == "e" else -1)
        return super().__new__(cls, (int(l), int(p)))

    l = property(lambda s: s[0])
    p = property(lambda s: s[1])
    dim = property(lambda s: 2 * s[0] + 1)

    def __str__(self):
        return f"{self[0]}{'e' if self[1] == 1 else 'o'}"

    __repr__ = __str__

    def is_scalar(self):
        return self[0] == 0 and self[1] == 1

    def __mul__(self, other):
        other = Irrep(other)
        return [Irrep(l, self[1] * other[1]) for l in range(abs(self[0] - other[0]), self[0] + other[0] + 1)]


class _MulIr(tuple):
    mul = property(lambda s: s[0])
    ir = property(lambda s: s[1])
    dim = property(lambda s: s[0] * s[1].dim)


class Irreps(tuple):
    def __new__(cls, spec=None):
        if isinstance(spec, Irreps):
            return spec
        items = []
        if isinstance(spec, str):
            for tok in spec.split("+"):
                tok = tok.strip()
                if not tok:
                    continue
                if "x" in tok:
                    mul, ir = tok.split("x")
                    items.append(_MulIr((int(mul), Irrep(ir))))
                else:
                    items.append(_MulIr((1, Irrep(tok))))
        elif spec is not None:
            for it in spec:
                mul, ir = it
                items.append(_MulIr((int(mul), Irrep(ir) if not isinstance(ir, tuple) or isinstance(ir, Irrep) else Irrep(*ir))))
        return super().__new__(cls, items)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, Irrep(l, p ** l)) for l in range(lmax + 1)])

    dim = property(lambda s: sum(mi.dim for mi in s))
    num_irreps = property(lambda s: sum(mi.mul for mi in s))

    def slices(self):
        out, i = [], 0
        for mi in self:
            out.append(slice(i, i + mi.dim))
            i += mi.dim
        return out

    def __str__(self):
        return "+".join(f"{mi.mul}x{mi.ir}" for mi in self)

    __repr__ = __str__


# --------------------------------------------------------------------------- spherical harmonics

SQRT3 = math.sqrt(3.0)
SQRT5 = math.sqrt(5.0)


def _unit(vec):
    return torch.nn.functional.normalize(vec, dim=-1)  # norm clamped at 1e-12, zero vector -> zero


def _y2_norm(u):
    """l=2 real SH polynomials of a unit vector, 'norm' normalised, e3nn order m=-2..2 (y polar)."""
    x, y, z = u[..., 0], u[..., 1], u[..., 2]
    return torch.stack([SQRT3 * x * z, SQRT3 * x * y, y * y - 0.5 * (x * x + z * z),
                        SQRT3 * y * z, (SQRT3 / 2.0) * (z * z - x * x)], dim=-1)


def spherical_harmonics(irreps, vec, normalize=True, normalization="component"):
    """[recalled] e3nn.o3.spherical_harmonics for l<=2: component normalisation = sqrt(2l+1) * norm-normalised."""
    assert normalization == "component"
    if isinstance(irreps, int):
        ls = [irreps]
    elif isinstance(irreps, str) and re.fullmatch(r"\s*\d+[eo]\s*", irreps):
        ls = [Irrep(irreps).l]
    else:
        ls = [mi.ir.l for mi in Irreps(irreps) for _ in range(mi.mul)]
    u = _unit(vec) if normalize else vec
    out = []
    for l in ls:
        if l == 0:
            out.append(torch.ones_like(u[..., :1]))
        elif l == 1:
            out.append(SQRT3 * u)
        elif l == 2:
            out.append(SQRT5 * _y2_norm(u))
        else:
            raise NotImplementedError("oracle restates spherical harmonics for l<=2 only")
    return torch.cat(out, dim=-1)


# --------------------------------------------------------------------------- tensor products


def _sym_traceless_from_y2(b):
    """3x3 symmetric traceless matrix M'(b) whose entries are linear in the five l=2 components b
    such that M'(Y2norm(v)) = v v^T - I/3 for a unit vector v (see _y2_norm)."""
    b0, b1, b2, b3, b4 = b.unbind(-1)
    mxz, mxy, myz = b0 / SQRT3, b1 / SQRT3, b3 / SQRT3
    myy = (2.0 / 3.0) * b2
    mzz = 0.5 * (-myy + (2.0 / SQRT3) * b4)
    mxx = 0.5 * (-myy - (2.0 / SQRT3) * b4)
    return torch.stack([torch.stack([mxx, mxy, mxz], -1),
                        torch.stack([mxy, myy, myz], -1),
                        torch.stack([mxz, myz, mzz], -1)], -2)


class FullTensorProduct(torch.nn.Module):
    """[recalled] o3.FullTensorProduct(sh(lmax=1), "2e"), SURVEY Appendix B.4.

    irreps_out (sorted) = 1x1o + 1x2e + 1x2o + 1x3o (20 components).  Down-stream (tor_bond_conv,
    whose node input has l<=1 and whose output is l=0) only the leading 1o block can couple, so only
    that block is computed; the remaining 17 components are returned as zeros (never read).
    1o block = sqrt(3) * sum_ij w3j(1,2,1)[i,j,k] a_i b_j with the Frobenius-normalised invariant
    tensor = (3/sqrt(10)) * M'(b) a, predicted sign '+' (Appendix B.4/D.7).
    """

    KAPPA = 3.0 / math.sqrt(10.0)

    def __init__(self, irreps_in1, irreps_in2):
        super().__init__()
        self.irreps_in1 = Irreps(irreps_in1)
        self.irreps_in2 = Irreps(irreps_in2)
        assert str(self.irreps_in1) == "1x0e+1x1o" and str(self.irreps_in2) == "1x2e", \
            "oracle restates FullTensorProduct(1x0e+1x1o, 2e) only (sh_lmax=1)"
        self.irreps_out = Irreps("1x1o+1x2e+1x2o+1x3o")

    def forward(self, x, y):
        a = x[..., 1:4]
        c = self.KAPPA * torch.einsum("...kj,...j->...k", _sym_traceless_from_y2(y), a)
        return torch.cat([c, c.new_zeros(c.shape[:-1] + (17,))], dim=-1)


class FullyConnectedTensorProduct(torch.nn.Module):
    """[recalled] o3.FullyConnectedTensorProduct(in, sh, out, shared_weights=False), SURVEY Appendix B.5,
    restricted to what the torsion heads need: `in` irreps with l<=1, `sh` whose first block is 1x1o and whose
    other blocks have l>=2, `out` irreps with l=0 only.

    Instructions in e3nn order (for in1, for in2, for out, kept if ir_out in ir1*ir2); 'uvw' weights
    [mul1, mul2, mul_out] flattened in instruction order; path weight sqrt(dim_out / sum_{paths->same out} mul1*mul2);
    w3j(1,1,0) = delta/sqrt(3).
    """

    def __init__(self, irreps_in1, irreps_in2, irreps_out, shared_weights=False):
        super().__init__()
        assert not shared_weights
        self.irreps_in1, self.irreps_in2, self.irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        assert all(mi.ir.l <= 1 for mi in self.irreps_in1) and all(mi.ir.l == 0 for mi in self.irreps_out)
        assert self.irreps_in2[0].mul == 1 and str(self.irreps_in2[0].ir) == "1o"
        assert all(mi.ir.l >= 2 for mi in self.irreps_in2[1:])
        self.instr = []
        for i1, m1 in enumerate(self.irreps_in1):
            for i2, m2 in enumerate(self.irreps_in2):
                for io, mo in enumerate(self.irreps_out):
                    if mo.ir in m1.ir * m2.ir:
                        self.instr.append((i1, i2, io))
        fan = {}
        for i1, i2, io in self.instr:
            fan[io] = fan.get(io, 0) + self.irreps_in1[i1].mul * self.irreps_in2[i2].mul
        self.path_weight = [math.sqrt(self.irreps_out[io].ir.dim / fan[io]) for (_, _, io) in self.instr]
        self.weight_numel = sum(self.irreps_in1[i1].mul * self.irreps_in2[i2].mul * self.irreps_out[io].mul
                                for i1, i2, io in self.instr)

    def forward(self, x, y, weight):
        sl1, slo = self.irreps_in1.slices(), self.irreps_out.slices()
        b = y[..., 0:3]
        out = x.new_zeros(x.shape[:-1] + (self.irreps_out.dim,))
        off = 0
        for (i1, i2, io), pw in zip(self.instr, self.path_weight):
            assert i2 == 0 and self.irreps_in1[i1].ir.l == 1
            mul1, mulo = self.irreps_in1[i1].mul, self.irreps_out[io].mul
            a = x[..., sl1[i1]].reshape(x.shape[:-1] + (mul1, 3))
            w = weight[..., off:off + mul1 * mulo].reshape(weight.shape[:-1] + (mul1, mulo))
            off += mul1 * mulo
            feat = (a * b.unsqueeze(-2)).sum(-1) / SQRT3          # [E, mul1]
            out[..., slo[io]] = out[..., slo[io]] + pw * torch.einsum("...u,...uw->...w", feat, w)
        assert off == self.weight_numel
        return out


class BatchNorm(torch.nn.Module):
    """[recalled] e3nn.nn.BatchNorm(irreps), eps=1e-5, affine, normalization='component', eval-mode forward
    (SURVEY Appendix B.2): scalar (0e) blocks are mean-shifted and biased; every block is scaled by
    weight / sqrt(running_var + eps), broadcast over the 2l+1 components."""

    def __init__(self, irreps, eps=1e-5):
        super().__init__()
        self.irreps = Irreps(irreps)
        ns = sum(mi.mul for mi in self.irreps if mi.ir.is_scalar())
        nf = self.irreps.num_irreps
        self.eps = eps
        self.register_buffer("running_mean", torch.zeros(ns))
        self.register_buffer("running_var", torch.ones(nf))
        self.weight = torch.nn.Parameter(torch.ones(nf))
        self.bias = torch.nn.Parameter(torch.zeros(ns))

    def forward(self, x):
        return batch_norm_eval(self.irreps, x, self.running_mean, self.running_var, self.weight, self.bias, self.eps)


def batch_norm_eval(irreps, x, running_mean, running_var, weight, bias, eps=1e-5):
    irreps = Irreps(irreps)
    out, ix, iw, ib = [], 0, 0, 0
    for mi in irreps:
        d = mi.ir.dim
        f = x[:, ix:ix + mi.mul * d].reshape(-1, mi.mul, d)
        ix += mi.mul * d
        if mi.ir.is_scalar():
            f = f - running_mean[ib:ib + mi.mul].reshape(1, -1, 1)
        f = f * (running_var[iw:iw + mi.mul] + eps).pow(-0.5).reshape(1, -1, 1)
        f = f * weight[iw:iw + mi.mul].reshape(1, -1, 1)
        if mi.ir.is_scalar():
            f = f + bias[ib:ib + mi.mul].reshape(1, -1, 1)
            ib += mi.mul
        iw += mi.mul
        out.append(f.reshape(-1, mi.mul * d))
    return torch.cat(out, dim=-1)


# --------------------------------------------------------------------------- scatter


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    """[recalled] torch_scatter.scatter for dim=0, reduce in {sum, mean}; mean divides by count.clamp(min=1)."""
    assert dim == 0
    n = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    out = src.new_zeros((n,) + tuple(src.shape[1:]))
    out.index_add_(0, index, src)
    if reduce in ("mean",):
        cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
        out = out / cnt.reshape((-1,) + (1,) * (src.dim() - 1))
    elif reduce not in ("sum", "add"):
        raise NotImplementedError(reduce)
    return out


def scatter_mean(src, index, dim=0, dim_size=None):
    return scatter(src, index, dim=dim, dim_size=dim_size, reduce="mean")


# --------------------------------------------------------------------------- neighbour search


def _graph_spans(batch, n):
    """[(graph id, start, stop)] of a SORTED batch vector (PyG batches are sorted)."""
    if batch is None:
        return [(0, 0, n)]
    assert bool((batch[1:] >= batch[:-1]).all()), "batch vector must be sorted"
    ids, counts = torch.unique_consecutive(batch, return_counts=True)
    stops = torch.cumsum(counts, 0).tolist()
    starts = [0] + stops[:-1]
    return list(zip(ids.tolist(), starts, stops))


TRUNCATION = "first_index"


def radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32, truncation=None):
    """[recalled] torch_cluster.radius: row0 indexes y (query), row1 indexes x;
    strict '<' on float32 squared distances; emitted in ascending (query, x) index order.  A query with more than
    `max_num_neighbors` matches keeps, by default ("first_index"), the FIRST max_num_neighbors in ascending x index: the
    CUDA kernel of torch_cluster 1.6.1 - what the reference executes on a GPU - scans x in index order and stops at the cap
    (its CPU path, nanoflann with unsorted results, keeps an implementation-defined subset).  "nearest" keeps the nearest
    ones and ties.  Evaluated graph by graph (memory-safe for 40 x 1111 atoms)."""
    rule = TRUNCATION if truncation is None else truncation
    assert rule in ("first_index", "nearest")
    sx = {g: (a, b) for g, a, b in _graph_spans(batch_x, x.shape[0])}
    qs, ns = [], []
    r2 = torch.as_tensor(r, dtype=x.dtype) ** 2
    for g, ya, yb in _graph_spans(batch_y, y.shape[0]):
        if g not in sx or yb == ya:
            continue
        xa, xb = sx[g]
        diff = y[ya:yb].unsqueeze(1) - x[xa:xb].unsqueeze(0)
        d2 = (diff * diff).sum(-1)
        ok = d2 < r2
        if ok.shape[1] > max_num_neighbors and int(ok.sum(1).max()) > max_num_neighbors:
            if rule == "nearest":
                d2m = torch.where(ok, d2, torch.full_like(d2, float("inf")))
                kth = torch.topk(d2m, max_num_neighbors, dim=1, largest=False).values[:, -1:]
                ok = ok & (d2m <= kth)
            else:
                ok = ok & (torch.cumsum(ok.long(), dim=1) <= max_num_neighbors)
        q, n = ok.nonzero(as_tuple=True)
        qs.append(q + ya)
        ns.append(n + xa)
    if not qs:
        return torch.zeros((2, 0), dtype=torch.long)
    return torch.stack([torch.cat(qs), torch.cat(ns)], 0)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target", truncation=None):
    """[recalled] torch_cluster.radius_graph: radius(x, x, r, max+1 if not loop) -> rows swapped to
    [neighbour; query], self loops dropped."""
    assert flow == "source_to_target"
    ei = radius(x, x, r, batch, batch, max_num_neighbors if loop else max_num_neighbors + 1, truncation=truncation)
    q, n = ei[0], ei[1]
    if not loop:
        keep = q != n
        q, n = q[keep], n[keep]
    return torch.stack([n, q], 0)


def knn_graph(x, k, batch=None, loop=False, flow="source_to_target"):
    """[recalled] torch_cluster.knn_graph: knn(x, x, k+1) -> [neighbour; query], self loops dropped; every
    query emits min(k, n_graph-1) edges ordered by increasing distance."""
    assert flow == "source_to_target" and not loop
    nbrs, qs = [], []
    for _, a, b in _graph_spans(batch, x.shape[0]):
        n = b - a
        if n <= 1:
            continue
        diff = x[a:b].unsqueeze(1) - x[a:b].unsqueeze(0)
        d2 = (diff * diff).sum(-1)
        d2.fill_diagonal_(float("inf"))
        kk = min(k, n - 1)
        idx = torch.topk(d2, kk, dim=1, largest=False, sorted=True).indices
        q = torch.arange(n).reshape(-1, 1).expand_as(idx)
        nbrs.append(idx.reshape(-1) + a)
        qs.append(q.reshape(-1) + a)
    if not nbrs:
        return torch.zeros((2, 0), dtype=torch.long)
    return torch.stack([torch.cat(nbrs), torch.cat(qs)], 0)
