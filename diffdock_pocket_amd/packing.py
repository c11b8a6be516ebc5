"""Host-side weight packing and shape descriptors for the fused conv kernel (csrc/ddp_conv.hip).

Runs once at `load_state_dict` time.  Turns the reference's parameter tensors
  fc.0.weight [hid, f_in], fc.0.bias [hid], fc.3.weight [weight_numel, hid], fc.3.bias [weight_numel]
  (reference models/score_model.py:100-105; weight_numel layout = blocks 0e,1o,1e,0o each row-major [U, n],
   models/layers.py:26-32,55-61)
into the tile-major, K-interleaved MFMA operand layout documented in ddp_conv.hip, with the 1/sqrt(U) of
models/layers.py:59 (or the e3nn path weight for the torsion heads) folded in.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import torch

from . import _lib as L


@dataclass
class BlockSpec:
    U: int
    n: int
    C: int
    out_off: int
    w_off: int                      # offset of this block inside the reference's flat per-edge weight vector
    scale: float                    # folded into the packed fc2 weight/bias
    segs: List[Tuple[int, int, int]]  # (kind, in_off, count)
    tile0: int = 0
    ntiles: int = 0
    nsub: int = 1
    ups: int = 1

    def finalize(self, tile0):
        if self.n > 32:
            self.nsub, self.ups = (self.n + 31) // 32, 1
            self.ntiles = self.U * self.nsub
        else:
            self.nsub, self.ups = 1, 32 // self.n
            self.ntiles = (self.U + self.ups - 1) // self.ups
        if self.C == 1 and self.ntiles % 2:
            self.ntiles += 1        # scalar blocks are processed in tile pairs; the extra tile is all zeros
        self.tile0 = tile0
        return tile0 + self.ntiles

    def column_rows(self):
        """LongTensor [ntiles*32]: row of fc.3.weight feeding packed column (tile, j); -1 = zero column."""
        t = torch.arange(self.ntiles).repeat_interleave(32)
        j = torch.arange(32).repeat(self.ntiles)
        if self.nsub > 1:
            u, sub = t // self.nsub, t % self.nsub
            ncol = sub * 32 + j
            valid = (ncol < self.n) & (u < self.U)
        else:
            us, ncol = j // self.n, j % self.n
            u = t * self.ups + us
            valid = (us < self.ups) & (u < self.U)
        rows = self.w_off + u * self.n + ncol
        return torch.where(valid, rows, torch.full_like(rows, -1))


@dataclass
class ConvSpec:
    f_in: int
    hid: int
    d_out: int
    weight_numel: int
    blocks: List[BlockSpec] = field(default_factory=list)

    def __post_init__(self):
        self.kp1 = (self.f_in + 7) // 8 * 8
        self.hp = (self.hid + 7) // 8 * 8
        self.hs = max(self.kp1, self.hp) + 4
        self.nct1 = (self.hid + 31) // 32
        t = 0
        for b in self.blocks:
            t = b.finalize(t)
        self.ntiles = t
        fb = max(64 * self.hs, 5 * 4096)      # staging tile / four per-wave partial regions + carry of phase 4
        for b in self.blocks:
            fb = max(fb, b.U * b.C * L.FS)
        self.fbuf_floats = (fb + 3) // 4 * 4
        if any(b.n > 64 or (b.C == 3 and b.n > 32) for b in self.blocks):
            raise NotImplementedError("HIP conv supports ns <= 64 and nv <= 32")

    def ctypes_shape(self) -> L.ConvShape:
        s = L.ConvShape()
        s.f_in, s.hid, s.kp1, s.hp, s.hs, s.nct1 = self.f_in, self.hid, self.kp1, self.hp, self.hs, self.nct1
        s.d_out, s.nblocks, s.fbuf_floats = self.d_out, len(self.blocks), self.fbuf_floats
        for i, b in enumerate(self.blocks):
            cb = s.blk[i]
            cb.U, cb.n, cb.C, cb.out_off = b.U, b.n, b.C, b.out_off
            cb.tile0, cb.ntiles, cb.nsub, cb.ups, cb.nseg = b.tile0, b.ntiles, b.nsub, b.ups, len(b.segs)
            for k, (kind, off, cnt) in enumerate(b.segs):
                cb.seg[k].kind, cb.seg[k].in_off, cb.seg[k].count = kind, off, cnt
        return s

    def flops_per_edge(self):
        """Algorithmic FLOPs per edge of the reference formulation."""
        c = sum(b.U * b.n * b.C for b in self.blocks)
        return 2 * self.f_in * self.hid + 2 * self.hid * self.weight_numel + 2 * c

    def mfma_flops_per_edge_executed(self):
        return 2 * self.kp1 * self.nct1 * 32 + 2 * self.hp * self.ntiles * 32


def irreps_muls(ns, nv, i):
    """(m0e, m1o, m1e, m0o) of irrep_seq[min(i,3)] (reference models/all_atom_score_model.py:95-100)."""
    i = min(i, 3)
    return (ns, nv if i >= 1 else 0, nv if i >= 2 else 0, ns if i >= 3 else 0)


def irreps_dim(m):
    return m[0] + 3 * m[1] + 3 * m[2] + m[3]


def faster_tp_spec(in_mul: Sequence[int], out_mul: Sequence[int], n_edge_features: int) -> ConvSpec:
    """Blocks of FasterTensorProduct (reference models/layers.py:26-31,40-53,82-85)."""
    m0e, m1o, m1e, m0o = in_mul
    n0e, n1o, n1e, n0o = out_mul
    o0e, o1o, o1e, o0o = 0, m0e, m0e + 3 * m1o, m0e + 3 * m1o + 3 * m1e          # input column offsets
    q0e, q1o, q1e, q0o = 0, n0e, n0e + 3 * n1o, n0e + 3 * n1o + 3 * n1e          # output column offsets
    table = [
        # (U, n, C, out_off, segs)
        (m0e + m1o, n0e, 1, q0e, [(L.F_SCALAR_S0, o0e, m0e), (L.F_DOT, o1o, m1o)]),
        (m0e + m1o + m1e, n1o, 3, q1o, [(L.F_SCALAR_S1, o0e, m0e), (L.F_VEC_S0, o1o, m1o), (L.F_CROSS, o1e, m1e)]),
        (m1o + m1e + m0o, n1e, 3, q1e, [(L.F_CROSS, o1o, m1o), (L.F_VEC_S0, o1e, m1e), (L.F_SCALAR_S1, o0o, m0o)]),
        (m1e + m0o, n0o, 1, q0o, [(L.F_DOT, o1e, m1e), (L.F_SCALAR_S0, o0o, m0o)]),
    ]
    blocks, w_off = [], 0
    for U, n, C, out_off, segs in table:
        if U * n > 0:
            blocks.append(BlockSpec(U=U, n=n, C=C, out_off=out_off, w_off=w_off, scale=1.0 / math.sqrt(U),
                                    segs=[s for s in segs if s[2] > 0]))
        w_off += U * n
    return ConvSpec(f_in=n_edge_features, hid=n_edge_features, d_out=irreps_dim(out_mul), weight_numel=w_off,
                    blocks=blocks)


def torsion_tp_spec(in_mul: Sequence[int], ns: int, n_edge_features: int) -> ConvSpec:
    """The two non-empty paths of o3.FullyConnectedTensorProduct(in, FullTensorProduct(sh,"2e").irreps_out,
    "ns x0o + ns x0e") (reference models/all_atom_score_model.py:194-202):
       (1o x sh.1o -> 0e) then (1e x sh.1o -> 0o); weights [m1o*ns | m1e*ns]; path weight 1/sqrt(mul_in);
       feature = dot(a_u, t)/sqrt(3).  Output layout [0o(ns) | 0e(ns)]."""
    m0e, m1o, m1e, m0o = in_mul
    o1o, o1e = m0e, m0e + 3 * m1o
    blocks, w_off = [], 0
    if m1o > 0:
        blocks.append(BlockSpec(U=m1o, n=ns, C=1, out_off=ns, w_off=w_off, scale=1.0 / math.sqrt(m1o),
                                segs=[(L.F_DOT, o1o, m1o)]))
        w_off += m1o * ns
    if m1e > 0:
        blocks.append(BlockSpec(U=m1e, n=ns, C=1, out_off=0, w_off=w_off, scale=1.0 / math.sqrt(m1e),
                                segs=[(L.F_DOT, o1e, m1e)]))
        w_off += m1e * ns
    if not blocks:
        raise NotImplementedError("torsion head needs at least one conv layer (1o node features)")
    return ConvSpec(f_in=n_edge_features, hid=n_edge_features, d_out=2 * ns, weight_numel=w_off, blocks=blocks)


def _pack_tiles(Wcols: torch.Tensor, kp: int) -> torch.Tensor:
    """Wcols [ncols (multiple of 32), K] -> flat float tensor in the layout
    w[((tile*(kp/8) + m)*2 + hh)*32 + j][i] = Wcols[tile*32 + j][8m + 4hh + i] (zero padded to kp)."""
    ncols, K = Wcols.shape
    assert ncols % 32 == 0 and kp % 8 == 0 and kp >= K
    W = torch.zeros(ncols, kp, dtype=torch.float32)
    W[:, :K] = Wcols
    W = W.reshape(ncols // 32, 32, kp // 8, 2, 4)          # [tile, j, m, hh, i]
    return W.permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


def pack_fc1(spec: ConvSpec, weight: torch.Tensor, bias: torch.Tensor):
    """fc.0: weight [hid, f_in] -> packed [nct1 tiles]; bias -> [nct1*32]."""
    hid, f_in = weight.shape
    assert (hid, f_in) == (spec.hid, spec.f_in)
    ncols = spec.nct1 * 32
    Wc = torch.zeros(ncols, f_in)
    Wc[:hid] = weight.detach().float().cpu()
    b = torch.zeros(ncols)
    b[:hid] = bias.detach().float().cpu()
    return _pack_tiles(Wc, spec.kp1), b


def pack_fc2(spec: ConvSpec, weight: torch.Tensor, bias: torch.Tensor):
    """fc.3: weight [weight_numel, hid] -> packed [ntiles tiles] with the block scale folded in."""
    wn, hid = weight.shape
    assert hid == spec.hid and wn == spec.weight_numel, (weight.shape, spec.weight_numel, spec.hid)
    weight = weight.detach().float().cpu()
    bias = bias.detach().float().cpu()
    cols, bcols = [], []
    for b in spec.blocks:
        rows = b.column_rows()
        valid = rows >= 0
        Wc = torch.zeros(rows.numel(), hid)
        Wc[valid] = weight[rows[valid]] * b.scale
        bc = torch.zeros(rows.numel())
        bc[valid] = bias[rows[valid]] * b.scale
        cols.append(Wc)
        bcols.append(bc)
    return _pack_tiles(torch.cat(cols, 0), spec.hp), torch.cat(bcols, 0)


def bn_affine(out_mul_blocks: Sequence[Tuple[int, int, bool]], running_mean, running_var, weight, bias, eps=1e-5):
    """e3nn BatchNorm (eval) as per-column scale/shift.
    out_mul_blocks: [(mul, dim, is_scalar_0e)] in output order.  Returns (scale[d_out], shift[d_out])."""
    scale_c = (weight.detach().double() / torch.sqrt(running_var.detach().double() + eps))
    scales, shifts, iw, ib = [], [], 0, 0
    for mul, dim, is_scalar in out_mul_blocks:
        s = scale_c[iw:iw + mul]
        scales.append(s.repeat_interleave(dim))
        if is_scalar:
            sh = bias.detach().double()[ib:ib + mul] - running_mean.detach().double()[ib:ib + mul] * s
            ib += mul
        else:
            sh = torch.zeros(mul, dtype=torch.float64)
        shifts.append(sh.repeat_interleave(dim))
        iw += mul
    return torch.cat(scales).float(), torch.cat(shifts).float()
