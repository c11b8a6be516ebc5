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
import os
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
    u_map: List[int] = None         # original feature index of each kept feature (None = identity)
    # source-node factorisation of the block's scalar-input segment (see faster_tp_spec(factorized=True))
    g_slot: int = -1                # -1 none, 0 = G array built from the 0e inputs, 1 = from the 0o inputs
    g_col0: int = 0                 # first column of this block inside a G row
    g_in_off: int = 0               # first column of the scalar inputs inside a node row
    g_count: int = 0                # number of scalar inputs (= rows of the factorised weight slab)
    g_u0: int = 0                   # original feature index of the first factorised feature

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
        if self.u_map is not None:
            um = torch.tensor(self.u_map + [0], dtype=torch.long)
            u = um[torch.clamp(u, max=len(self.u_map))]
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
        self.g_cols = [0, 0]
        for b in self.blocks:
            if b.g_slot >= 0:
                self.g_cols[b.g_slot] = max(self.g_cols[b.g_slot], b.g_col0 + b.n)
        self.factorized = any(b.g_slot >= 0 for b in self.blocks)
        if self.factorized:
            self.fbuf_floats += (64 * max(self.g_cols) + 3) // 4 * 4   # + the tv[64][g_cols] region of the factorised part
        if any(b.n > 64 or (b.C == 3 and b.n > 32) for b in self.blocks):
            raise NotImplementedError("HIP conv supports ns <= 64 and nv <= 32")
        self.roles, self.nrounds = (conv32_roles(self.blocks) if self.factorized else ([[] for _ in range(CONV32_WAVES)], 0))

    def ctypes_shape(self) -> L.ConvShape:
        s = L.ConvShape()
        s.f_in, s.hid, s.kp1, s.hp, s.hs, s.nct1 = self.f_in, self.hid, self.kp1, self.hp, self.hs, self.nct1
        s.d_out, s.nblocks, s.fbuf_floats = self.d_out, len(self.blocks), self.fbuf_floats
        s.g_cols[0], s.g_cols[1] = self.g_cols
        s.nrounds = self.nrounds
        for w, segs in enumerate(self.roles):
            s.nrole[w] = len(segs)
            for k, (bi, t0, st, cnt, rnd) in enumerate(segs):
                r = s.role[w][k]
                r.block, r.tile0, r.tstride, r.count, r.round = bi, t0, st, cnt, rnd
        for i, b in enumerate(self.blocks):
            cb = s.blk[i]
            cb.U, cb.n, cb.C, cb.out_off = b.U, b.n, b.C, b.out_off
            cb.tile0, cb.ntiles, cb.nsub, cb.ups, cb.nseg = b.tile0, b.ntiles, b.nsub, b.ups, len(b.segs)
            cb.g_slot, cb.g_col0 = b.g_slot, b.g_col0
            for k, (kind, off, cnt) in enumerate(b.segs):
                cb.seg[k].kind, cb.seg[k].in_off, cb.seg[k].count = kind, off, cnt
        return s

    def flops_per_edge(self):
        """Algorithmic FLOPs per edge of the reference formulation."""
        c = sum(b.U * b.n * b.C for b in self.blocks)
        return 2 * self.f_in * self.hid + 2 * self.hid * self.weight_numel + 2 * c

    def mfma_flops_per_edge_executed(self):
        """fp32 MFMA FLOPs per edge the tile loops issue (K and columns padded to the 32-column / 8-k tiles)."""
        return 2 * self.kp1 * self.nct1 * 32 + 2 * self.hp * self.ntiles * 32

    def fc_flops_per_edge(self):
        """The part of useful_flops_per_edge that is the two dense fc products (fc1 + the kept fc2 columns): what runs as fp16 hi/lo
        split products (three matrix-core FLOPs per product FLOP) in the h2 form of the kernels."""
        return 2 * self.f_in * self.hid + 2 * self.hid * sum(b.U * b.n for b in self.blocks)

    def useful_flops_per_edge(self):
        """USEFUL fp32 FLOPs per edge of the formulation the kernel executes, without any padding: fc1, the fc2 columns of
        the features that stay on the per-edge path (all of them on the direct path, the vector-input ones on the
        factorised path), their contraction with the basis features, and the per-edge G contraction h @ G[src] of the
        factorised features (2 * hid * g_cols).  This is what roofline.frac in bench.py counts."""
        cols = sum(b.U * b.n for b in self.blocks)
        contr = sum(b.U * b.n * b.C for b in self.blocks)
        return 2 * self.f_in * self.hid + 2 * self.hid * cols + 2 * contr + 2 * self.hid * sum(self.g_cols)


# ------------------------------------------------------------------------------------------------ 32-edge kernel roles
CONV32_WAVES, MAX_ROLE_SEGS = 4, 2


def conv32_roles(blocks: Sequence["BlockSpec"]):
    """Static work split of the factorised (32-edge workgroup) kernel: which tiles each of its 4 waves runs.

    An item is a set of tiles whose results land in the SAME output columns: a whole block (n <= 32: its tiles differ only
    in the features u they cover) or one 32-column half of a block (n > 32: tiles u*2 + sub).  A wave that owns a whole item
    accumulates it in registers with no cross-wave reduction at all.  To balance the four matrix pipes up to two items are cut
    in two (by feature range); the two parts are then summed in LDS in a fixed order (part 0 in round 0, part 1 in round 1).
    A wave holds at most MAX_ROLE_SEGS parts and at most 4 result components (16 registers each) at a time.  Loads are
    balanced in tile COST: a vector tile (three contraction components, lane-group sums) measured ~1.3 x a scalar tile on the
    3dpf launches (in-kernel stamps), and with equal tile counts the waves holding the vector blocks finished ~20 % late.
    Returns ([per wave: [(block index, first tile, tile stride, tile count, round)]], number of rounds)."""
    import itertools
    COST = {1: 10, 3: int(os.environ.get("DDP_ROLE_COST3", "13"))}     # (diagnostic override: the vector tile's relative cost)
    items = []     # (block index, first tile, stride, count, C)
    for bi, b in enumerate(blocks):
        if b.ntiles == 0:
            continue
        if b.nsub > 1:
            for sub in range(b.nsub):
                items.append((bi, sub, b.nsub, b.U, b.C))
        else:
            items.append((bi, 0, 1, b.ntiles, b.C))
    if not items:
        return [[] for _ in range(CONV32_WAVES)], 0
    total = sum(it[3] * COST[it[4]] for it in items)
    best = None

    def assign(parts):
        """parts: [(bi, t0, stride, count, C, round)] -> best (max load, waves) with <= 2 parts and <= 4 components per wave."""
        n = len(parts)
        if n > CONV32_WAVES * MAX_ROLE_SEGS:
            return None
        order = sorted(range(n), key=lambda i: -parts[i][3] * COST[parts[i][4]])
        res = [None]

        def rec(k, waves, loads, comps):
            if res[0] is not None and max(loads) >= res[0][0]:
                return
            if k == n:
                res[0] = (max(loads), [list(w) for w in waves])
                return
            i = order[k]
            seen = set()
            for w in range(CONV32_WAVES):
                key = (loads[w], comps[w], len(waves[w]))
                if key in seen or len(waves[w]) >= MAX_ROLE_SEGS or comps[w] + parts[i][4] > 4:
                    continue
                seen.add(key)
                waves[w].append(i)
                loads[w] += parts[i][3] * COST[parts[i][4]]
                comps[w] += parts[i][4]
                rec(k + 1, waves, loads, comps)
                waves[w].pop()
                loads[w] -= parts[i][3] * COST[parts[i][4]]
                comps[w] -= parts[i][4]

        rec(0, [[] for _ in range(CONV32_WAVES)], [0] * CONV32_WAVES, [0] * CONV32_WAVES)
        return res[0]

    ideal = -(-total // CONV32_WAVES)
    for nsplit in (0, 1, 2):
        for which in itertools.combinations(range(len(items)), nsplit):
            cuts = [range(1, items[i][3]) for i in which]
            for cut in itertools.product(*cuts):
                parts = []
                for i, it in enumerate(items):
                    if i in which:
                        c = cut[which.index(i)]
                        bi, t0, st, cnt, C = it
                        parts.append((bi, t0, st, c, C, 0))
                        parts.append((bi, t0 + c * st, st, cnt - c, C, 1))
                    else:
                        parts.append(it + (0,))
                got = assign(parts)
                if got is None:
                    continue
                score = (got[0], nsplit)
                if best is None or score < best[0]:
                    best = (score, parts, got[1])
                if got[0] <= ideal + 5:
                    break
            if best is not None and best[0][0] <= ideal + 5:
                break
        if best is not None and best[0][0] <= ideal + 5:
            break
    if best is None:
        raise NotImplementedError("no role assignment for the 32-edge conv kernel (too many weight blocks with tiles)")
    _, parts, waves = best
    roles = [[(parts[i][0], parts[i][1], parts[i][2], parts[i][3], parts[i][5]) for i in sorted(w, key=lambda i: (parts[i][4], parts[i][0]))]
             for w in waves]
    nrounds = 1 + max(p[5] for p in parts)
    return roles, nrounds


def irreps_muls(ns, nv, i):
    """(m0e, m1o, m1e, m0o) of irrep_seq[min(i,3)] (reference models/all_atom_score_model.py:95-100)."""
    i = min(i, 3)
    return (ns, nv if i >= 1 else 0, nv if i >= 2 else 0, ns if i >= 3 else 0)


def irreps_dim(m):
    return m[0] + 3 * m[1] + 3 * m[2] + m[3]


def faster_tp_spec(in_mul: Sequence[int], out_mul: Sequence[int], n_edge_features: int, factorized: bool = False) -> ConvSpec:
    """Blocks of FasterTensorProduct (reference models/layers.py:26-31,40-53,82-85).

    factorized=True: the basis features built from SCALAR inputs (a0e*s0, a0e*s1, a0o*s1, a0o*s0) depend on the
    edge only through sh, so their part of the message is  sh-factor(e) * sum_k h[e,k] * G[src(e)][k, n]  with the
    per-SOURCE-NODE tensor  G[j][k, n] = sum_u a[j,u] * W2[(u,n), k] / sqrt(U)   (exact algebra, fp32).  Those features
    (84 % of the fc.3 weight at ns=60) leave the per-edge MFMA work; G is produced by one plain GEMM per conv
    (`factor_weights` below gives its right-hand side).  Only the vector-input features stay in `segs`/`u_map`."""
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
    gcol = [0, 0]
    for U, n, C, out_off, segs in table:
        if U * n > 0:
            segs = [s for s in segs if s[2] > 0]
            blk = BlockSpec(U=U, n=n, C=C, out_off=out_off, w_off=w_off, scale=1.0 / math.sqrt(U), segs=segs)
            if factorized:
                keep, umap, u0 = [], [], 0
                for kind, off, cnt in segs:
                    if kind in (L.F_SCALAR_S0, L.F_SCALAR_S1):
                        slot = 0 if off == o0e else 1
                        blk.g_slot, blk.g_col0, blk.g_in_off, blk.g_count, blk.g_u0 = slot, gcol[slot], off, cnt, u0
                        gcol[slot] += n
                    else:
                        keep.append((kind, off, cnt))
                        umap += list(range(u0, u0 + cnt))
                    u0 += cnt
                blk.segs, blk.u_map, blk.U = keep, umap, len(umap)
            blocks.append(blk)
        w_off += U * n
    return ConvSpec(f_in=n_edge_features, hid=n_edge_features, d_out=irreps_dim(out_mul), weight_numel=w_off,
                    blocks=blocks)


def factor_weights(spec: ConvSpec, weight: torch.Tensor, bias: torch.Tensor):
    """Right-hand sides of the per-source-node GEMMs of a factorised conv, per G slot s (0: 0e inputs, 1: 0o inputs):
         Wg[s]  [n_in, DDP_G_LD]          row j of x[:, in_off:in_off+n_in] @ Wg[s] = [G[j] (hg/4, g_cols, 4) | Gb[j] (g_cols) | 0]
         Bg[s]  [n_in, g_cols[s]]         Gb[j] = x[j, in_off:in_off+n_in] @ Bg[s]     (the fc.3 bias part; also inside Wg)
       hg = hid rounded up to 4 (zero rows); the k index of G is interleaved in quads, G[j][k/4][c][k%4], so that a lane
       of the kernel's G pass fetches 4 consecutive k of its column with one 16-byte load (include/ddp_hip.h),
       with the block scale 1/sqrt(U_orig) folded in.  Returns ([Wg0, Wg1], [Bg0, Bg1], [in_off0, in_off1]); entries
       of unused slots are None."""
    weight = weight.detach().float().cpu()
    bias = bias.detach().float().cpu()
    Wg, Bg, offs = [None, None], [None, None], [0, 0]
    for slot in (0, 1):
        blks = [b for b in spec.blocks if b.g_slot == slot]
        if not blks:
            continue
        n_in = blks[0].g_count
        assert all(b.g_count == n_in and b.g_in_off == blks[0].g_in_off for b in blks)
        W = torch.zeros(n_in, spec.hid, spec.g_cols[slot])
        Bm = torch.zeros(n_in, spec.g_cols[slot])
        for b in blks:
            rows = b.w_off + (b.g_u0 + torch.arange(n_in)).reshape(-1, 1) * b.n + torch.arange(b.n).reshape(1, -1)
            W[:, :, b.g_col0:b.g_col0 + b.n] = (weight[rows] * b.scale).permute(0, 2, 1)     # [u, n, hid] -> [u, hid, n]
            Bm[:, b.g_col0:b.g_col0 + b.n] = bias[rows] * b.scale
        hg = (spec.hid + 3) // 4 * 4
        Wq = torch.zeros(n_in, hg, spec.g_cols[slot])
        Wq[:, :spec.hid] = W
        Wq = Wq.reshape(n_in, hg // 4, 4, spec.g_cols[slot]).permute(0, 1, 3, 2)            # [u, k/4, c, k%4]
        # one right-hand side per slot: [G part | Gb part | zero padding to DDP_G_LD] - the bias columns ride in the same product
        gld = ((hg + 1) * spec.g_cols[slot] + 31) // 32 * 32
        Wfull = torch.zeros(n_in, gld)
        Wfull[:, :hg * spec.g_cols[slot]] = Wq.reshape(n_in, -1)
        Wfull[:, hg * spec.g_cols[slot]:(hg + 1) * spec.g_cols[slot]] = Bm
        Wg[slot], Bg[slot], offs[slot] = Wfull.contiguous(), Bm.contiguous(), blks[0].g_in_off
    return Wg, Bg, offs


def split_bf16x3(W: torch.Tensor) -> torch.Tensor:
    """Stage-A weights [nb, K, ncols] fp32 as three bfloat16 terms w = hi + mid + lo (hi = bf16(w), mid = bf16(w - hi),
    lo = bf16(w - hi - mid): 24 significant bits, the residuals are exact in fp32), in the operand order of
    v_mfma_f32_32x32x16_bf16: [nb][plane][k/16][k/8 % 2][ncols][8] (include/ddp_hip.h, ddp_stage_a `w_bf16x3`), K zero-padded
    to a multiple of 16."""
    nb, K, ncols = W.shape
    KP = (K + 15) // 16 * 16
    Wf = torch.zeros((nb, KP, ncols), dtype=torch.float32, device=W.device)
    Wf[:, :K] = W.float()
    hi = Wf.to(torch.bfloat16)
    r1 = Wf - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    planes = torch.stack([hi, mid, lo], dim=1)                                  # [nb, 3, KP, ncols]
    return planes.reshape(nb, 3, KP // 16, 2, 8, ncols).permute(0, 1, 2, 3, 5, 4).contiguous()


def split_h2(W: torch.Tensor, unified_scale: float = 0.0) -> torch.Tensor:
    """Stage-A weights [nb, K, ncols] fp32 as two fp16 planes w = hi + lo / 2048 (hi = fp16(w), lo = fp16((w - hi) * 2048)) in the
    operand order of v_mfma_f32_32x32x16_f16: [nb][plane][k/16][k/8 % 2][ncols][8] (include/ddp_hip.h, ddp_stage_a_h2 `w_h2`), K
    zero-padded to a multiple of 16.  unified_scale = S > 0: UNIFIED planes of S w instead (lo = fp16(S w - hi), both halves at one
    scale): ddp_stage_a_gh's form, S = 1 / DDP_GH_SX (GH_SW below)."""
    nb, K, ncols = W.shape
    KP = (K + 15) // 16 * 16
    Wf = torch.zeros((nb, KP, ncols), dtype=torch.float32, device=W.device)
    Wf[:, :K] = W.float() * (unified_scale if unified_scale > 0.0 else 1.0)
    hi = Wf.to(torch.float16)
    lo = ((Wf - hi.float()) * (1.0 if unified_scale > 0.0 else H2_SCALE)).to(torch.float16)
    if not bool(torch.isfinite(hi).all()):
        raise NotImplementedError("stage-A weight outside the fp16 range (|w| > 65504): the fp16 hi/lo form cannot represent it")
    planes = torch.stack([hi, lo], dim=1)                                       # [nb, 2, KP, ncols]
    return planes.reshape(nb, 2, KP // 16, 2, 8, ncols).permute(0, 1, 2, 3, 5, 4).contiguous()


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


H2_SCALE = 2048.0     # csrc/ddp_conv.hip DDP_H2_SCALE
# plane scales of ddp_conv_rows' unified hi/lo planes (include/ddp_hip.h DDP_ROWS_S*): edge_attr_, weights, h, G
ROWS_SX, ROWS_SW, ROWS_SH, ROWS_SG = 16.0, 256.0, 16.0, 32.0
GH_SW = 1.0 / 2.0       # ddp_stage_a_gh: planes of w / DDP_GH_SX (the kernel splits x at DDP_GH_SX = 2: typical |16 w| and |2 x| keep normal lo halves)


def h2_steps(spec: "ConvSpec") -> int:
    """k16 steps of the fp16 hi/lo split ("h2") form of a conv's fc products, 0 if the shape has none: f_in = hid = 3 ns with ns
    one of the released architectures' multiplicities (the size classes of csrc/ddp_conv.hip, H2Class)."""
    if spec.f_in != spec.hid:
        return 0
    return {180: 12, 96: 6, 72: 5, 48: 3}.get(spec.hid, 0)


def _pack_tiles_h2(Wcols: torch.Tensor, ns16: int, unified_scale: float = 0.0) -> torch.Tensor:
    """Wcols [ncols (multiple of 32), K] fp32 -> fp16 operand planes of v_mfma_f32_32x32x16_f16's B operand, v = hi + lo / 2048
    (hi = fp16(v), lo = fp16((v - hi) * 2048)):  [tile][ks][plane][hh][j][8] = plane(W)[tile*32 + j][16 ks + 8 hh + i], K zero-padded
    to 16 * ns16 (include/ddp_hip.h, ddp_conv_task_t::w1h / w2h).
    unified_scale = S > 0: the UNIFIED planes of ddp_conv_rows instead, V = S v = hi + lo (lo = fp16(V - hi), both halves at one scale)."""
    ncols, K = Wcols.shape
    kp = 16 * ns16
    assert ncols % 32 == 0 and kp >= K
    W = torch.zeros(ncols, kp, dtype=torch.float32)
    W[:, :K] = Wcols
    if unified_scale > 0.0:
        W = W * unified_scale
    hi = W.to(torch.float16)
    lo = ((W - hi.float()) * (1.0 if unified_scale > 0.0 else H2_SCALE)).to(torch.float16)
    if not bool(torch.isfinite(hi).all()):
        raise NotImplementedError("fc weight outside the fp16 range (|w| > 65504, ddp_conv_rows' planes: |w| > 255): the fp16 hi/lo form cannot "
                                  "represent it")
    Pl = torch.stack([hi, lo], 0).reshape(2, ncols // 32, 32, ns16, 2, 8)      # [plane, tile, j, ks, hh, i]
    return Pl.permute(1, 3, 0, 4, 2, 5).contiguous().reshape(-1)


def rows16_pos(ncols: int) -> torch.Tensor:
    """LongTensor [ncols] (ncols a multiple of 32): DDP_ROWS16_POS per 32-column tile - where column c = 32 t + j of fc.0's output sits in the
    stream of ddp_conv_rows' 16x16x32 form (rows_form 1): position 32 t + 16 ((j & 7) >> 2) + 4 (j >> 3) + (j & 3).  The transposed fc1
    product then leaves lane (edge, g) with the h columns 32 t + 8 g + i, i < 8, as its A fragment of k32 step t: natural k order."""
    c = torch.arange(ncols)
    j = c % 32
    return (c - j) + 16 * ((j & 7) >> 2) + 4 * (j >> 3) + (j & 3)


def _pack_tiles_16(Wcols: torch.Tensor, ns16: int, unified_scale: float) -> torch.Tensor:
    """Wcols [ncols (multiple of 32), K] fp32 -> unified fp16 hi/lo planes of S w in the operand images of v_mfma_f32_16x16x32_f16
    (ddp_conv_task_t::rows_form = 1): per 32-column tile 2 ns16 fragments of 1 KiB ordered [k32 step s][column tile ct][plane], each
    [k group g < 4][column n < 16][8 halves] = plane(W)[column 16 ct + n][k = 32 s + 8 g + i], K zero-padded to 16 ns16."""
    ncols, K = Wcols.shape
    kp = 16 * ns16
    assert ncols % 32 == 0 and kp >= K and ns16 % 2 == 0
    W = torch.zeros(ncols, kp, dtype=torch.float32)
    W[:, :K] = Wcols
    W = W * unified_scale
    hi = W.to(torch.float16)
    lo = (W - hi.float()).to(torch.float16)
    if not bool(torch.isfinite(hi).all()):
        raise NotImplementedError("fc weight outside the fp16 range of ddp_conv_rows' planes (|w| > 255): the fp16 hi/lo form cannot represent it")
    Pl = torch.stack([hi, lo], 0).reshape(2, ncols // 32, 2, 16, ns16 // 2, 4, 8)      # [plane, tile, ct, n, s, g, i]
    return Pl.permute(1, 4, 2, 0, 5, 3, 6).contiguous().reshape(-1)                    # [tile, s, ct, plane, g, n, i]


def pack_fc1_h2(spec: ConvSpec, weight: torch.Tensor):
    """fc.0 weight [hid, f_in] as fp16 hi/lo planes per 32-column tile (bias: pack_fc1's)."""
    hid, f_in = weight.shape
    Wc = torch.zeros(spec.nct1 * 32, f_in)
    Wc[:hid] = weight.detach().float().cpu()
    return _pack_tiles_h2(Wc, h2_steps(spec))


def pack_fc2_h2(spec: ConvSpec, weight: torch.Tensor):
    """fc.3 weight [weight_numel, hid] as fp16 hi/lo planes per tile, block scale folded in BEFORE the split (bias: pack_fc2's)."""
    weight = weight.detach().float().cpu()
    cols = []
    for b in spec.blocks:
        rows = b.column_rows()
        valid = rows >= 0
        Wc = torch.zeros(rows.numel(), spec.hid)
        Wc[valid] = weight[rows[valid]] * b.scale
        cols.append(Wc)
    return _pack_tiles_h2(torch.cat(cols, 0), h2_steps(spec))


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

# ------------------------------------------------------------------------------------------------ row-stationary kernel (ddp_conv_rows)
def spec_form1_chunks(spec: "ConvSpec") -> bool:
    """Feature chunks exist in the 16x16x32 form of the rows kernel only, which is also the only one the direct convs run through."""
    return not spec.factorized


def rows_bias_in_k(spec: "ConvSpec") -> bool:
    """ddp_conv_task_t::rows_bias_k: a DIRECT conv (every feature a stream tile: hundreds of tiles, whose bias table alone would take the
    LDS of two workgroups) carries the fc.3 bias in the padding k row `hid` of its tiles - possible where hid is not a multiple of 16."""
    return (not spec.factorized) and spec.hid % 16 != 0


def rows_supported(spec: "ConvSpec") -> bool:
    """Shapes ddp_conv_rows runs: convs of the size classes ns = 60 / 32 (f_in = hid = 180 / 96: the README's large and small score
    models) whose per-wave feature rows and bias table fit the kernel's LDS plan - the factorised convs, and (round 6, rows_form 1 only) the
    direct ones where the bias can ride in k (rows_bias_in_k: hid = 180)."""
    ns16 = h2_steps(spec)
    if ns16 not in (12, 6) or (not spec.factorized and not rows_bias_in_k(spec)):
        return False
    # the host entry's block checks (csrc/ddp_conv_rows.hip, ddp_conv_rows): shapes it would refuse keep the 32-edge kernel instead of
    # failing at launch
    for b in spec.blocks:
        if b.C not in (1, 3) or b.n < 1 or b.n > 64 or (b.C == 3 and b.n > 32):
            return False
        if b.nsub < 1 or b.nsub > 2 or b.ups < 1 or (b.nsub > 1) != (b.n > 32) or (b.nsub == 1 and b.ups != 32 // b.n):
            return False
    frows = max([b.U * b.C for b in spec.blocks if b.U > 0] + [0])
    if spec_form1_chunks(spec):
        frows = min(frows, 72)          # (csrc/ddp_conv_rows16.hip R16_FROWS: larger blocks build their features in chunks; rows_form 1 only)
    priv = max((frows * 36 * 4 + 127) // 128 * 128 + 1408 + 2048, ns16 * 1024)
    nts = spec.nct1 if rows_bias_in_k(spec) else spec.nct1 + sum(len(t) for _, _, t in rows_segments(spec))     # (tiles of the bias table)
    # two 4-wave workgroups per CU: ring of one tile's pieces + bias table + four private areas each
    return 2 * (2 * ns16 * 1024 + (nts * 128 + 127) // 128 * 128 + 4 * ((priv + 127) // 128 * 128)) <= 160 * 1024


def rows_kperm(ns16: int) -> torch.Tensor:
    """LongTensor [16 ns16]: slot (ks, hh, i) of an A / B operand fragment of ddp_conv_rows -> the h column (k index of fc.3 / G) it holds,
    DDP_ROWS_KPERM of include/ddp_hip.h.  The kernel computes h = relu(fc1) as the TRANSPOSED product (A = fc.0 tile, B = edge_attr_),
    whose 32 x 32 accumulator tile ct leaves lane (edge r, hh) with the h columns 32 ct + (j & 3) + 8 (j >> 2) + 4 hh, j < 16, of its own
    edge: registers j = 0..7 are the fragment ks = 2 ct, j = 8..15 the fragment ks = 2 ct + 1 - no transpose through LDS."""
    ks = torch.arange(ns16).reshape(-1, 1, 1)
    hh = torch.arange(2).reshape(1, -1, 1)
    i = torch.arange(8).reshape(1, 1, -1)
    j = 8 * (ks % 2) + i
    return (32 * (ks // 2) + (j % 4) + 8 * (j // 4) + 4 * hh).reshape(-1)


def rows_segments(spec: "ConvSpec"):
    """The output-column segments ddp_conv_rows walks, in order: (block index, 32-column part, [packed tile index of the part's stream
    tiles in feature order]).  A block with n > 32 has ceil(n / 32) parts (tiles u * nsub + part), one part otherwise; blocks whose
    features are all factorised have parts without stream tiles."""
    segs = []
    for bi, b in enumerate(spec.blocks):
        nparts = (b.n + 31) // 32
        for part in range(nparts):
            if b.U == 0:
                tiles = []
            elif b.nsub > 1:
                tiles = [b.tile0 + u * b.nsub + part for u in range(b.U)]
            else:
                tiles = [b.tile0 + t for t in range((b.U + b.ups - 1) // b.ups)]
            segs.append((bi, part, tiles))
    return segs


def rows_split_segments(spec: "ConvSpec", nsplit: int):
    """[(seg0, seg1, fc.3 tiles)]: the shape's output segments (rows_segments order) cut into at most nsplit contiguous ranges, the largest
    range as small as possible (ddp_conv_task_t::rows_seg0 / rows_seg1: one conv spread over several workgroups per 128 edges)."""
    import itertools
    counts = [len(t) for _, _, t in rows_segments(spec)]
    n = len(counts)
    nsplit = max(1, min(nsplit, n))
    best = None
    for cuts in itertools.combinations(range(1, n), nsplit - 1):
        edges = (0,) + cuts + (n,)
        sums = [sum(counts[a:b]) for a, b in zip(edges[:-1], edges[1:])]
        if best is None or max(sums) < best[0]:
            best = (max(sums), edges, sums)
    _, edges, sums = best
    return [(a, b, c) for a, b, c in zip(edges[:-1], edges[1:], sums)]


def rows_stream(spec: "ConvSpec", w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, form: int = 0, bias_in_k: bool = False,
                seg_range=None):
    """(wsh, bsp) of ddp_conv_task_t for ddp_conv_rows: fc.0's nct1 column tiles (natural k order: their K is edge_attr_), then the fc.3
    tiles of `spec` (block scale folded in) segment by segment with the k index permuted by rows_kperm; UNIFIED fp16 hi/lo planes of
    ROWS_SW w per tile (_pack_tiles_h2), the bias words fp32 [tiles, 32] at the scale of their tile's accumulator (fc.0: ROWS_SW ROWS_SX,
    fc.3: ROWS_SH ROWS_SW).
    form = 1 (ddp_conv_task_t::rows_form, csrc/ddp_conv_rows16.hip): the same stream in the operand images of v_mfma_f32_16x16x32_f16
    (_pack_tiles_16): fc.0's output columns placed by rows16_pos inside every 32-column tile (bias words in the same positions), the k of
    the fc.3 tiles in natural order.
    bias_in_k (ddp_conv_task_t::rows_bias_k, form 1): fc.0 gets an output column `hid` with zero weights and bias 1 (h[hid] = relu(1) = 1),
    every fc.3 tile its bias as k row `hid`; the bias words of the fc.3 tiles are zero.
    seg_range = (seg0, seg1): fc.0's tiles followed by the tiles of those segments only (ddp_conv_task_t::rows_seg0 / rows_seg1)."""
    ns16 = h2_steps(spec)
    assert ns16 > 0
    hid, f_in = w1.shape
    assert not bias_in_k or (form == 1 and hid < 16 * ns16 and hid < spec.nct1 * 32)
    W1c = torch.zeros(spec.nct1 * 32, f_in)
    W1c[:hid] = w1.detach().float().cpu()
    b1c = torch.zeros(spec.nct1 * 32)
    b1c[:hid] = b1.detach().float().cpu() * (ROWS_SW * ROWS_SX)
    if bias_in_k:
        b1c[hid] = 1.0 * (ROWS_SW * ROWS_SX)
    if form == 1:
        pos = rows16_pos(spec.nct1 * 32)
        W1p, b1p = torch.zeros_like(W1c), torch.zeros_like(b1c)
        W1p[pos] = W1c
        b1p[pos] = b1c
        W1c, b1c = W1p, b1p
        t1 = _pack_tiles_16(W1c, ns16, ROWS_SW).reshape(spec.nct1, -1)
    else:
        t1 = _pack_tiles_h2(W1c, ns16, ROWS_SW).reshape(spec.nct1, -1)
    w2 = w2.detach().float().cpu()
    b2 = b2.detach().float().cpu()
    cols, bcols = [], []
    for b in spec.blocks:
        rows = b.column_rows()
        valid = rows >= 0
        Wc = torch.zeros(rows.numel(), spec.hid)
        Wc[valid] = w2[rows[valid]] * b.scale
        bc = torch.zeros(rows.numel())
        bc[valid] = b2[rows[valid]] * b.scale
        cols.append(Wc)
        bcols.append(bc)
    segs = rows_segments(spec)
    if seg_range is not None:
        segs = segs[seg_range[0]:seg_range[1]]
    order = [t for _, _, tiles in segs for t in tiles]
    if order:
        Wall = torch.cat(cols, 0)                                   # [ntiles * 32, hid]
        Wp = torch.zeros(Wall.shape[0], 16 * ns16)
        Wp[:, :spec.hid] = Wall
        if bias_in_k:
            Wp[:, spec.hid] = torch.cat(bcols, 0)
        if form == 1:
            t2 = _pack_tiles_16(Wp, ns16, ROWS_SW).reshape(Wall.shape[0] // 32, -1)[order]      # (natural k)
        else:
            Wp = Wp[:, rows_kperm(ns16)]                                # fragment slot -> permuted k
            t2 = _pack_tiles_h2(Wp, ns16, ROWS_SW).reshape(Wall.shape[0] // 32, -1)[order]
        bs2 = torch.cat(bcols, 0).reshape(-1, 32)[order] * (0.0 if bias_in_k else ROWS_SH * ROWS_SW)
        return torch.cat([t1, t2], 0).reshape(-1).contiguous(), torch.cat([b1c.reshape(-1, 32), bs2], 0).contiguous()
    return t1.reshape(-1).contiguous(), b1c.reshape(-1, 32).contiguous()


def gh_parts(spec: "ConvSpec", slot: int):
    """The column parts of G array `slot` in plane form (ddp_conv_task_t::gh), in layout order: [(block index, part, G column of the
    part's first column, width, width rounded up to 4, padded columns in front)]."""
    parts, cum = [], 0
    for bi, b in enumerate(spec.blocks):
        if b.g_slot != slot:
            continue
        for part in range((b.n + 31) // 32):
            w = min(32, b.n - 32 * part)
            wp = (w + 3) // 4 * 4
            parts.append((bi, part, b.g_col0 + 32 * part, w, wp, cum))
            cum += wp
    return parts


def gh_ld(hid: int, gcp: int) -> int:
    """DDP_GH_LD: floats per node of a G array in plane form with gcp padded columns."""
    return (((hid + 7) // 8 * 8 + 1) * gcp + 31) // 32 * 32


GH3_FP32_COLS = (0, 1, 4, 5, 2, 6)     # plane form 1: product column (inside its 8-column group) of the j-th fp32 value a group stores


def gh3_ld(hid: int, gcp: int) -> int:
    """DDP_GH3_LD: floats per node of a G array in plane form 1 (fp16 hi + a continuation byte: 24 bytes per 8 values; Gb as 6 fp32 values
    per 24-byte group; whole 384-byte pieces = 16 groups = the four column blocks of a stage-A wave)."""
    n8 = (hid + 7) // 8
    return (n8 * gcp + (gcp + 5) // 6 + 15) // 16 * 96


def factor_weights_gh(spec: "ConvSpec", weight: torch.Tensor, bias: torch.Tensor, fmt: int = 0, form: int = 0):
    """factor_weights for ddp_conv_rows: right-hand sides whose product columns are ordered [part][k8 group][column c of the part][8 k's
    of the group] (gh_parts; the k's of group g = 2 ks + hh are rows_kperm's slots (ks, hh, 0..7); h columns >= hid and the padding
    columns of a part are zero), then Gb per padded column, then zero padding to DDP_GH_LD.  ddp_stage_a_gh writes the groups of a row
    as unified fp16 hi/lo planes, every part a contiguous tile [k8][c][plane][8] (ddp_conv_task_t::gh): the plane scale ROWS_SG rides in
    the G columns, the accumulator scale ROWS_SH ROWS_SG in the Gb columns.
    fmt = 1 (plane form 1, ddp_stage_a_gh3): the same plane groups, then Gb in groups of SIX values per 8 product columns (the j-th value
    of a group at product column GH3_FP32_COLS[j]: stage A stores those six, in that order), then zero padding to a multiple of 128 columns;
    the output row is gh3_ld = 6 columns / 8 floats.
    Returns ([Wg0, Wg1], [in_off0, in_off1], [part widths of slot 0, of slot 1])."""
    ns16 = h2_steps(spec)
    assert ns16 > 0
    weight = weight.detach().float().cpu()
    bias = bias.detach().float().cpu()
    n8 = (spec.hid + 7) // 8
    # (rows_form 1: h, and with it G's k, is in natural order - the same bytes per node, other k's in the groups)
    kp = torch.arange(8 * n8) if form == 1 else rows_kperm(ns16)[:8 * n8]
    Wg, offs, widths = [None, None], [0, 0], [None, None]
    for slot in (0, 1):
        blks = [b for b in spec.blocks if b.g_slot == slot]
        if not blks:
            continue
        gc = spec.g_cols[slot]
        n_in = blks[0].g_count
        assert all(b.g_count == n_in and b.g_in_off == blks[0].g_in_off for b in blks)
        W = torch.zeros(n_in, 16 * ns16, gc)
        Bm = torch.zeros(n_in, gc)
        for b in blks:
            rows = b.w_off + (b.g_u0 + torch.arange(n_in)).reshape(-1, 1) * b.n + torch.arange(b.n).reshape(1, -1)
            W[:, :spec.hid, b.g_col0:b.g_col0 + b.n] = (weight[rows] * b.scale).permute(0, 2, 1)     # [u, n, hid] -> [u, hid, n]
            Bm[:, b.g_col0:b.g_col0 + b.n] = bias[rows] * b.scale
        parts = gh_parts(spec, slot)
        gcp = sum(p[4] for p in parts)
        ld = gh_ld(spec.hid, gcp) if fmt != 1 else gh3_ld(spec.hid, gcp) // 6 * 8
        Wfull = torch.zeros(n_in, ld)
        Wk = W[:, kp].reshape(n_in, n8, 8, gc)                                                       # [u, k8, i, column]
        for _, _, c0, w, wp, cum in parts:
            tile = torch.zeros(n_in, n8, wp, 8)
            tile[:, :, :w] = Wk[:, :, :, c0:c0 + w].permute(0, 1, 3, 2) * ROWS_SG
            Wfull[:, 8 * n8 * cum:8 * n8 * (cum + wp)] = tile.reshape(n_in, -1)
            if fmt != 1:
                Wfull[:, 8 * n8 * gcp + cum:8 * n8 * gcp + cum + w] = Bm[:, c0:c0 + w] * (ROWS_SH * ROWS_SG)
            else:
                cpad = cum + torch.arange(w)                                                         # padded column of the slot
                Wfull[:, 8 * n8 * gcp + 8 * (cpad // 6) + torch.tensor(GH3_FP32_COLS)[cpad % 6]] = Bm[:, c0:c0 + w] * (ROWS_SH * ROWS_SG)
        Wg[slot], offs[slot], widths[slot] = Wfull.contiguous(), blks[0].g_in_off, [p[4] for p in parts]
    return Wg, offs, widths


def gh_dest_table(widths: Sequence[int], n8: int, ncols: int, fmt: int = 0) -> torch.Tensor:
    """int32 [ncols / 8, 2] for ddp_stage_a_gh: float offsets inside a G row of the two 16-byte pieces of every 8-column group of the
    product factor_weights_gh sets up; bit 0 of entry [g][0] set = a plane group (its values leave as fp16 hi / lo words).  Group
    g = n8 cum_p + k8 w_p + c of part p (cum_p = the widths in front of it) is the k8 group of column c: its hi piece sits at 16-byte unit
    2 (n8 cum_p + k8 w_p + c), its lo piece right behind it - so the pieces of four neighbouring columns (one 32-column block of the
    product) fill one 128-byte line.  Groups behind the parts (Gb, padding) are 8 fp32 columns at their own place."""
    ng = ncols // 8
    tab = torch.empty((ng, 2), dtype=torch.int32)
    g = torch.arange(ng, dtype=torch.int64)
    if fmt == 1:
        # plane form 1: EVERY group g of the product sits at byte 24 g of the row (ddp_stage_a_gh3 reads bit 0 of [g][0] only: a plane group;
        # the offsets are written for the record: 6 g floats, the group's 8 bytes behind its 16)
        gcp = sum(widths)
        assert 8 * (n8 * gcp + (gcp + 5) // 6) <= ncols and ncols % 128 == 0
        tab[:, 0] = (6 * g).int()
        tab[:, 1] = (6 * g + 4).int()
        tab[:n8 * gcp, 0] += 1
        return tab
    tab[:, 0] = (8 * g).int()
    tab[:, 1] = (8 * g + 4).int()
    cum = 0
    for w in widths:
        gs = n8 * cum
        gl = torch.arange(n8 * w, dtype=torch.int64)
        unit = 2 * (gs + gl)
        tab[gs:gs + n8 * w, 0] = (4 * unit + 1).int()
        tab[gs:gs + n8 * w, 1] = (4 * (unit + 1)).int()
        cum += w
    assert 8 * n8 * cum + cum <= ncols
    return tab
