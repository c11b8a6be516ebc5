"""MI355X-native drop-in for the reference's all-atom `TensorProductScoreModel`.

Same constructor signature and `forward(data) -> (tr_pred, rot_pred, tor_pred, sc_tor_pred)` as
reference models/all_atom_score_model.py:21-436, same `state_dict()` key layout so
reference checkpoints load with `load_state_dict`.  What runs where:

  host-side PyTorch on the ROCm device (north-star: graph construction stays in PyTorch)
      neighbour search (graph.py), node encoders (embedding sums + one Linear), sinusoidal embeddings,
      the [B,3]-sized tr/rot magnitude MLPs and table lookups
  hand-written HIP through the C ABI of libddp_hip.so (include/ddp_hip.h)
      edge featurisation (RBF + MLP + spherical harmonics)            ddp_edge_featurize
      fused fc -> tensor product -> message for the 9 convs / layer   ddp_conv_messages
      segmented mean + e3nn BatchNorm + residual                      ddp_segment_reduce
      torsion-head harmonics                                          ddp_torsion_sh

There is no eager/CPU fallback: `forward` raises if the inputs are not on a HIP device or the library is missing.
Configurations outside the README models (sh_lmax != 1, second-order irreps, odd_parity, separate or asynchronous noise
schedules, affinity prediction, parallel > 1) raise NotImplementedError instead of silently differing; smooth_edges is supported
(cosine edge weights folded into the edge harmonics, engine._front).
`confidence_mode=True` builds the confidence model (same convs, scalar-mean + MLP head).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from . import _lib as L
from . import graph as G
from . import launch as K
from . import packing as P
from .synthetic import LIG_FEATURE_DIMS, REC_ATOM_FEATURE_DIMS, REC_RESIDUE_FEATURE_DIMS

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


# ------------------------------------------------------------------------------------------------ parameter holders
class GaussianSmearing(nn.Module):
    """Parameter holder + host formula of reference models/score_model.py:661-671."""

    def __init__(self, start=0.0, stop=5.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer("offset", offset)


class AtomEncoder(nn.Module):
    """reference models/score_model.py:54-82 (host-side PyTorch: embedding sums + one Linear)."""

    def __init__(self, emb_dim, feature_dims, sigma_embed_dim, lm_embedding_type=None):
        super().__init__()
        self.num_categorical_features = len(feature_dims)
        if lm_embedding_type is not None and lm_embedding_type != "esm":
            raise ValueError("LM Embedding type was not correctly determined. LM embedding type: ", lm_embedding_type)
        lm_dim = 1280 if lm_embedding_type == "esm" else 0
        self.additional_features_dim = sigma_embed_dim + lm_dim
        self.atom_embedding_list = nn.ModuleList()
        for dim in feature_dims:
            emb = nn.Embedding(dim, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)
        if self.additional_features_dim > 0:
            self.additional_features_embedder = nn.Linear(self.additional_features_dim + emb_dim, emb_dim)

    def forward(self, x_cat, extra):
        emb = 0
        for i in range(self.num_categorical_features):
            emb = emb + self.atom_embedding_list[i](x_cat[:, i].long())
        if self.additional_features_dim > 0:
            emb = self.additional_features_embedder(torch.cat([emb, extra], dim=1))
        return emb

    def forward_split(self, x_cat, static_extra, sigma_emb, cached):
        """Same value as forward(x_cat, cat([static_extra, sigma_emb])) with the Linear split by input columns: the part
        fed by the node's own features (embedding sums, ESM block: 1280 of the 1340+ input columns of the receptor
        encoder) does not change between denoising steps and is kept through `cached` (score_model._cached); only the
        sigma-embedding columns are multiplied per call."""
        lin = self.additional_features_embedder
        n_static = lin.in_features - sigma_emb.shape[1]

        def static():
            emb = 0
            for i in range(self.num_categorical_features):
                emb = emb + self.atom_embedding_list[i](x_cat[:, i].long())
            feats = emb if static_extra is None else torch.cat([emb, static_extra.float()], dim=1)
            return torch.nn.functional.linear(feats, lin.weight[:, :n_static], lin.bias)

        return cached(static) + sigma_emb @ lin.weight[:, n_static:].t()


class OldAtomEncoder(nn.Module):
    """reference models/score_model.py:17-52 (legacy slicing kept literally)."""

    def __init__(self, emb_dim, feature_dims, sigma_embed_dim, lm_embedding_type=None):
        super().__init__()
        self.num_categorical_features = len(feature_dims)
        self.num_scalar_features = sigma_embed_dim
        self.lm_embedding_type = lm_embedding_type
        self.atom_embedding_list = nn.ModuleList()
        for dim in feature_dims:
            emb = nn.Embedding(dim, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)
        if self.num_scalar_features > 0:
            self.linear = nn.Linear(self.num_scalar_features, emb_dim)
        if lm_embedding_type is not None:
            if lm_embedding_type != "esm":
                raise ValueError("LM Embedding type was not correctly determined. LM embedding type: ", lm_embedding_type)
            self.lm_embedding_dim = 1280
            self.lm_embedding_layer = nn.Linear(self.lm_embedding_dim + emb_dim, emb_dim)

    def forward(self, x_cat, extra):
        emb = 0
        for i in range(self.num_categorical_features):
            emb = emb + self.atom_embedding_list[i](x_cat[:, i].long())
        if self.num_scalar_features > 0:
            emb = emb + self.linear(extra[:, :self.num_scalar_features])
        if self.lm_embedding_type is not None:
            emb = self.lm_embedding_layer(torch.cat([emb, extra[:, -self.lm_embedding_dim:]], dim=1))
        return emb


class IrrepsBatchNorm(nn.Module):
    """Parameter holder with e3nn.nn.BatchNorm's state_dict layout."""

    def __init__(self, blocks):  # blocks: [(mul, dim, is_0e)]
        super().__init__()
        self.blocks = list(blocks)
        n_scalar = sum(m for m, _, s in blocks if s)
        n_feat = sum(m for m, _, _ in blocks)
        self.register_buffer("running_mean", torch.zeros(n_scalar))
        self.register_buffer("running_var", torch.ones(n_feat))
        self.weight = nn.Parameter(torch.ones(n_feat))
        self.bias = nn.Parameter(torch.zeros(n_scalar))


def _mlp(n_in, n_hidden, n_out, dropout):
    return nn.Sequential(nn.Linear(n_in, n_hidden), nn.ReLU(), nn.Dropout(dropout), nn.Linear(n_hidden, n_out))


class _PackedConv:
    __slots__ = ("w1p", "b1p", "w2p", "b2p", "bn_scale", "bn_shift", "wg", "bg", "g_in_off", "w1h", "w2h",
                 "wsh", "bsp", "wgh", "gh_groups", "gh_ld", "gh_fmt", "rows_form", "rows_bias_k", "rows_seg", "rows_nts")     # (the last seven: ddp_conv_rows' weight stream and stage-A right-hand sides)


G_PLANES3_DEFAULT = "1"      # model.g_planes3 unless DDP_G_PLANES3 says otherwise (round 6, late: the 19-bit form is the default)
DIRECT_ROWS_DEFAULT = "1"    # model.direct_rows unless DDP_DIRECT_ROWS says otherwise (round 6, late)
ROWS_MFMA16_DEFAULT = "1"    # model.rows_mfma16 unless DDP_ROWS_MFMA16 says otherwise (round 6: 15.5 against 16.1 ms per 40-sample step)


class TensorProductConvLayer(nn.Module):
    """Parameter layout of reference models/score_model.py:84-107; the arithmetic of its forward (:108-125) runs in
    the HIP kernels.  `forward` keeps the reference call signature for a single conv (used by the parity tests);
    the score model itself batches the nine convs of a layer into one launch."""

    def __init__(self, spec: P.ConvSpec, out_blocks, batch_norm=True, dropout=0.0, spec_g: Optional[P.ConvSpec] = None):
        super().__init__()
        self.spec = spec
        self.spec_g = spec_g          # source-node factorised variant of the same conv (None: not available)
        self.out_blocks = list(out_blocks)
        self.fc = _mlp(spec.f_in, spec.hid, spec.weight_numel, dropout)
        self.batch_norm = IrrepsBatchNorm(out_blocks) if batch_norm else None
        self._packed: Optional[_PackedConv] = None
        self._packed_g: Optional[_PackedConv] = None
        self._packed_d = None           # (packed(), its copy with the direct conv's weight stream): packed_rows_direct
        # which forms of the row-stationary kernel a layer runs through: a model sets these on its layers (rows_mfma16, g_planes3,
        # direct_rows); a stand-alone layer takes the same defaults
        self.rows_form = 1 if os.environ.get("DDP_ROWS_MFMA16", ROWS_MFMA16_DEFAULT) == "1" else 0
        self.gh_fmt = 1 if os.environ.get("DDP_G_PLANES3", G_PLANES3_DEFAULT) == "1" else 0
        self.direct_rows = os.environ.get("DDP_DIRECT_ROWS", DIRECT_ROWS_DEFAULT) == "1"

    def packed_g(self, device) -> _PackedConv:
        """Weights for the factorised path: fc.3 tiles of the vector-input features only + the GEMM right-hand sides
        that turn source-node scalars into G / Gb (packing.factor_weights)."""
        if self._packed_g is None or self._packed_g.w1p.device != device:
            base = self.packed(device)
            pk = _PackedConv()
            pk.w1p, pk.b1p, pk.bn_scale, pk.bn_shift, pk.w1h = base.w1p, base.b1p, base.bn_scale, base.bn_shift, base.w1h
            w2p, b2p = P.pack_fc2(self.spec_g, self.fc[3].weight, self.fc[3].bias)
            if w2p.numel() == 0:
                w2p, b2p = torch.zeros(64), torch.zeros(32)
            pk.w2p, pk.b2p = w2p.to(device), b2p.to(device)
            pk.w2h = None
            if P.h2_steps(self.spec_g) > 0:
                w2h = P.pack_fc2_h2(self.spec_g, self.fc[3].weight)
                pk.w2h = (w2h if w2h.numel() else torch.zeros(64, dtype=torch.float16)).to(device)
            wg, bg, offs = P.factor_weights(self.spec_g, self.fc[3].weight, self.fc[3].bias)
            pk.wg = [w.to(device) if w is not None else None for w in wg]
            pk.bg = [b.to(device) if b is not None else None for b in bg]
            pk.g_in_off = offs
            # the 128-edge row-stationary kernel (ddp_conv_rows; size classes ns = 60 / 32): the fc.0 / fc.3 tiles as one stream in the
            # kernel's k order, and stage-A right-hand sides whose product ddp_stage_a_gh writes as fp16 hi/lo planes
            pk.wsh = pk.bsp = pk.wgh = pk.gh_groups = pk.gh_ld = None
            pk.rows_bias_k, pk.rows_seg, pk.rows_nts = 0, (0, 0), 0
            pk.gh_fmt = int(getattr(self, "gh_fmt", 0))     # plane form of G (ddp_conv_task_t::gh_fmt; set by the model: g_planes3)
            pk.rows_form = int(getattr(self, "rows_form", 0))   # operand images of the rows kernel (ddp_conv_task_t::rows_form: rows_mfma16)
            if P.rows_supported(self.spec_g):
                try:
                    wsh, bsp = P.rows_stream(self.spec_g, self.fc[0].weight, self.fc[0].bias, self.fc[3].weight, self.fc[3].bias, form=pk.rows_form)
                except NotImplementedError:
                    wsh = None      # a weight beyond the unified planes' range (|w| > 255): this conv keeps the 32-edge kernel
                if wsh is not None:
                    pk.wsh, pk.bsp = wsh.to(device), bsp.to(device)
                    wgh, _, widths = P.factor_weights_gh(self.spec_g, self.fc[3].weight, self.fc[3].bias, fmt=pk.gh_fmt, form=pk.rows_form)
                    pk.wgh = [w.to(device) if w is not None else None for w in wgh]
                    pk.gh_groups = widths          # (per slot: the padded widths of the G array's column parts)
                    # floats per node of the G array stage A writes (plane form 1: shorter than the product's columns)
                    pk.gh_ld = [None if w is None else (P.gh3_ld(self.spec_g.hid, sum(ws)) if pk.gh_fmt == 1 else w.shape[1])
                                for w, ws in zip(wgh, widths)]
            self._packed_g = pk
        return self._packed_g

    def node_tensors(self, pk: _PackedConv, x_src: torch.Tensor, rows: bool = False):
        """Stage A of the factorised conv: per-source-node rows [G | Gb | pad] = x_scalar @ Wg (ddp_stage_a); rows: in the plane
        form ddp_conv_rows reads (ddp_stage_a_gh)."""
        lib = L.load()
        g = [None, None]
        N = x_src.shape[0]
        if rows:
            for slot in (0, 1):
                if pk.wgh[slot] is None:
                    continue
                w = pk.wgh[slot]
                wh = P.split_h2(w.unsqueeze(0), unified_scale=P.GH_SW)
                g[slot] = torch.empty((N, pk.gh_ld[slot]), device=x_src.device, dtype=torch.float32)
                offs = (C.c_int32 * 1)(pk.g_in_off[slot])
                dest = P.gh_dest_table(pk.gh_groups[slot], (self.spec_g.hid + 7) // 8, w.shape[1], fmt=pk.gh_fmt).to(x_src.device)
                fn = lib.ddp_stage_a_gh3 if pk.gh_fmt == 1 else lib.ddp_stage_a_gh
                L.check(fn(x_src.data_ptr(), x_src.shape[1], N, None, None, N, offs, 1, w.data_ptr(), wh.data_ptr(), w.shape[0],
                           w.shape[1], g[slot].data_ptr(), pk.gh_ld[slot], None, dest.data_ptr(), _stream()), "ddp_stage_a_gh")
            return g
        for slot in (0, 1):
            if pk.wg[slot] is None:
                continue
            w = pk.wg[slot]
            g[slot] = torch.empty((N, w.shape[1]), device=x_src.device, dtype=torch.float32)
            offs = (C.c_int32 * 1)(pk.g_in_off[slot])
            L.check(lib.ddp_stage_a(x_src.data_ptr(), x_src.shape[1], N, None, None, N, offs, 1, w.data_ptr(), None, w.shape[0], w.shape[1],
                                    g[slot].data_ptr(), w.shape[1], _stream()), "ddp_stage_a")
        return g

    def packed(self, device) -> _PackedConv:
        if self._packed is None or self._packed.w1p.device != device:
            pk = _PackedConv()
            w1p, b1p = P.pack_fc1(self.spec, self.fc[0].weight, self.fc[0].bias)
            w2p, b2p = P.pack_fc2(self.spec, self.fc[3].weight, self.fc[3].bias)
            if self.batch_norm is not None:
                bn = self.batch_norm
                sc, sh = P.bn_affine(self.out_blocks, bn.running_mean.cpu(), bn.running_var.cpu(), bn.weight.cpu(), bn.bias.cpu())
            else:
                sc, sh = torch.ones(self.spec.d_out), torch.zeros(self.spec.d_out)
            pk.w1p, pk.b1p, pk.w2p, pk.b2p = (t.to(device) for t in (w1p, b1p, w2p, b2p))
            pk.bn_scale, pk.bn_shift = sc.to(device), sh.to(device)
            # the same weights as fp16 hi/lo operand planes: the fc products then run on the fp16 matrix cores (ddp_conv.hip, h2 form)
            pk.w1h = pk.w2h = None
            if P.h2_steps(self.spec) > 0:
                pk.w1h = P.pack_fc1_h2(self.spec, self.fc[0].weight).to(device)
                pk.w2h = P.pack_fc2_h2(self.spec, self.fc[3].weight).to(device)
            pk.wsh = pk.bsp = pk.wgh = pk.gh_groups = pk.gh_ld = None
            pk.gh_fmt, pk.rows_form, pk.rows_bias_k, pk.rows_seg, pk.rows_nts = 0, 0, 0, (0, 0), 0
            self._packed = pk
            self._packed_d = None
        return self._packed

    def packed_rows_direct(self, device, nsplit: int = 1):
        """The DIRECT conv through the row-stationary kernel (round 6: rows_form 1, the fc.3 bias in the padding k row - packing.rows_stream(
        bias_in_k)): packed() plus the weight stream, built on first use (8 MB per conv: only the convs that run direct get one).  nsplit > 1:
        a LIST of packs, one per range of output segments (packing.rows_split_segments; each with fc.0's tiles and its own segments' tiles:
        ddp_conv_task_t::rows_seg0 / rows_seg1 / rows_nts) - the tasks of one conv spread over several workgroups per 128 edges; nsplit = 1:
        the one pack.  None where the option is off, the shape is not one of the kernel's, or a weight / bias lies beyond the unified planes'
        range (ddp_conv_messages then)."""
        base = self.packed(device)
        if not (bool(getattr(self, "direct_rows", False)) and int(getattr(self, "rows_form", 0)) == 1 and not self.spec.factorized
                and P.rows_supported(self.spec)):
            return None
        if getattr(self, "_packed_d", None) is None or self._packed_d[0] is not base:
            self._packed_d = (base, {})
        cache = self._packed_d[1]
        ranges = P.rows_split_segments(self.spec, nsplit) if nsplit > 1 else [None]
        key = len(ranges)
        if key not in cache:
            pks = []
            for rg in ranges:
                pk = _PackedConv()
                for name in ("w1p", "b1p", "w2p", "b2p", "bn_scale", "bn_shift", "w1h", "w2h"):
                    setattr(pk, name, getattr(base, name))
                pk.wsh = pk.bsp = pk.wgh = pk.gh_groups = pk.gh_ld = None
                pk.gh_fmt, pk.rows_form, pk.rows_bias_k = 0, 1, 1
                pk.rows_seg, pk.rows_nts = ((rg[0], rg[1]), self.spec.nct1 + rg[2]) if rg is not None else ((0, 0), 0)
                try:
                    wsh, bsp = P.rows_stream(self.spec, self.fc[0].weight, self.fc[0].bias, self.fc[3].weight, self.fc[3].bias, form=1, bias_in_k=True,
                                             seg_range=None if rg is None else (rg[0], rg[1]))
                    pk.wsh, pk.bsp = wsh.to(device), bsp.to(device)
                    pks.append(pk)
                except NotImplementedError:
                    pks = None
                    break
            cache[key] = pks
        pks = cache[key]
        if pks is None:
            return None
        return pks if nsplit > 1 else pks[0]

    def forward(self, node_attr, edge_index, edge_attr, edge_sh, out_nodes=None, reduce="mean", edge_weight=1.0,
                factorized=False):
        if reduce != "mean" or not (isinstance(edge_weight, (int, float)) and edge_weight == 1.0):
            raise NotImplementedError("HIP conv implements reduce='mean', edge_weight=1")
        if edge_index.numel() == 0:
            return torch.tensor(0, dtype=node_attr.dtype, device=node_attr.device)
        _require_hip(node_attr)
        dev = node_attr.device
        n_out = int(out_nodes) if out_nodes is not None else node_attr.shape[0]
        csr = G.build_csr(edge_index[0].long(), edge_index[1].long(), n_out)
        x = node_attr.float().contiguous()
        ea = edge_attr.float().contiguous()
        sh = edge_sh.float().contiguous()
        if sh.shape[1] != 4:
            raise NotImplementedError("edge_sh must be [E,4] (lmax=1) or pre-contracted torsion harmonics [0,t]")
        msg = torch.empty((csr.n_edges, self.spec.d_out), device=dev, dtype=torch.float32)
        if factorized:
            if self.spec_g is None:
                raise NotImplementedError("this conv has no factorised variant")
            pk = self.packed_g(dev)
            so = G.source_order(csr, x.shape[0])
            rows = K.rows_mode(pk) and ea.shape[1] % 12 == 0
            g = self.node_tensors(pk, x, rows=rows)
            if rows:     # (ddp_conv_rows gathers edge_attr_ as three segments of ns columns: here three column ranges of one array)
                w3 = ea.shape[1] // 3
                segs = [(ea[:, i * w3:], so.eid, ea.shape[1], w3) for i in range(3)]
            else:
                segs = [(ea, so.eid, ea.shape[1], ea.shape[1])]
            task = _make_task(pk, x, x.shape[1], so, sh, segs, msg, g=g, rows=rows)
            _launch_convs(self.spec_g, [task], flops_spec=self.spec)
        else:
            pkr = self.packed_rows_direct(dev) if ea.shape[1] % 12 == 0 else None      # (the row-stationary kernel's direct form: direct_rows)
            if pkr is not None and K.rows_mode(pkr):
                w3 = ea.shape[1] // 3
                task = _make_task(pkr, x, x.shape[1], csr, sh, [(ea[:, i * w3:], csr.eid, ea.shape[1], w3) for i in range(3)], msg, rows=True)
            else:
                task = _make_task(self.packed(dev), x, x.shape[1], csr, sh, [(ea, csr.eid, ea.shape[1], ea.shape[1])], msg)
            _launch_convs(self.spec, [task])
        out = torch.zeros((n_out, self.spec.d_out), device=dev, dtype=torch.float32)
        _launch_reduce(out, self.spec.d_out, n_out, self.spec.d_out, [(msg, csr, self.packed(dev))], accumulate=False)
        return out


# ------------------------------------------------------------------------------------------------ launch helpers
# (diffdock_pocket_amd/launch.py; the names below are kept for the tests and tools that import them from here)
_require_hip, _stream, _ptr = K.require_hip, K.stream, K.ptr
ConvProfiler, SectionTimer, set_conv_profiler = K.ConvProfiler, K.SectionTimer, K.set_conv_profiler
_EdgeMLPPack, _edge_featurize = K.EdgeMLPPack, K.edge_featurize
_make_task, _launch_convs, _launch_reduce = K.make_task, K.launch_convs, K.launch_reduce


class _EncoderPack:
    """Weights of one node encoder laid out for ddp_node_linear (csrc/ddp_node.hip): the embedding tables stacked row-wise
    with their first rows, the Linear weights transposed to [K, ns].  AtomEncoder (models/score_model.py:54-82) is ONE job
    [emb_sum | ESM | sigma_emb] @ W^T + b; OldAtomEncoder (:17-52, legacy slicing kept literally: with an ESM block its
    "scalar features" are the first sigma_embed_dim ESM columns and its "ESM block" the last 1280 columns of
    [ESM | sigma_emb]) is a job `emb_sum + scalars @ W_s^T + b_s` followed, with ESM, by a second one over
    [stage 1 | ESM[sd:] | sigma_emb]."""

    def __init__(self, enc, dev):
        tabs = [e.weight.detach().float() for e in enc.atom_embedding_list]
        self.table = torch.cat(tabs, 0).contiguous().to(dev)
        self.feat_off = [0]
        for t in tabs[:-1]:
            self.feat_off.append(self.feat_off[-1] + t.shape[0])
        self.n_cat, self.emb_dim = len(tabs), tabs[0].shape[1]
        self.old = isinstance(enc, OldAtomEncoder)
        if self.old:
            self.has_lm = enc.lm_embedding_type is not None
            self.n_scalar = enc.num_scalar_features
            if self.n_scalar > 0:
                self.w1, self.b1 = enc.linear.weight.detach().float().t().contiguous().to(dev), enc.linear.bias.detach().float().to(dev)
            if self.has_lm:
                lin = enc.lm_embedding_layer
                self.w2, self.b2 = lin.weight.detach().float().t().contiguous().to(dev), lin.bias.detach().float().to(dev)
        else:
            self.has_extra = enc.additional_features_dim > 0
            if self.has_extra:
                lin = enc.additional_features_embedder
                self.w, self.b = lin.weight.detach().float().t().contiguous().to(dev), lin.bias.detach().float().to(dev)


def _node_job(n_rows, out, ncols, w, bias, zero_to=0, cat=None, pack=None, emb_mode=0, dense=(), sigma=None, sig_out=None, add=None):
    """One ddp_node_job_t.  dense: [(tensor [n, ld] float32 with unit column stride, first column, width)]; sigma:
    None | ("t", t [n] (any stride), scale, freq [sd/2], sd) | ("emb", tensor [n, sd])."""
    j = L.NodeJob()
    j.n_rows, j.out, j.ld_out, j.ncols, j.zero_to = n_rows, out.data_ptr(), out.stride(0), ncols, zero_to
    j.w, j.bias = w.data_ptr(), (bias.data_ptr() if bias is not None else 0)
    if cat is not None and emb_mode:
        j.cat, j.ld_cat, j.n_cat, j.table, j.emb_dim, j.emb_mode = cat.data_ptr(), cat.stride(0), pack.n_cat, pack.table.data_ptr(), pack.emb_dim, emb_mode
        for f, o in enumerate(pack.feat_off):
            j.feat_off[f] = o
    for d, (ten, c0, n) in enumerate(dense):
        j.dense[d], j.ld_dense[d], j.n_dense[d] = ten.data_ptr() + 4 * c0, ten.stride(0), n
    if sigma is not None and sigma[0] == "t":
        _, t, scale, freq, sd = sigma
        j.t, j.t_stride, j.scale, j.freq, j.sd = t.data_ptr(), (t.stride(0) if t.numel() > 1 else 0), scale, freq.data_ptr(), sd
    elif sigma is not None:
        j.sig_emb, j.ld_sig, j.sd = sigma[1].data_ptr(), sigma[1].stride(0), sigma[1].shape[1]
    if sig_out is not None:
        j.sig_out, j.ld_sig_out = sig_out.data_ptr(), sig_out.stride(0)
    if add is not None:
        j.add, j.ld_add = add.data_ptr(), add.stride(0)
    return j


def _launch_node_jobs(jobs):
    lib = L.load()
    for i in range(0, len(jobs), L.DDP_MAX_NODE_JOBS):
        part = jobs[i:i + L.DDP_MAX_NODE_JOBS]
        arr = (L.NodeJob * len(part))(*part)
        L.check(lib.ddp_node_linear(arr, len(part), _stream()), "ddp_node_linear")


# ------------------------------------------------------------------------------------------------ the model
class TensorProductScoreModel(nn.Module):
    def __init__(self, t_to_sigma, device, timestep_emb_func, in_lig_edge_features=4, sigma_embed_dim=32, sh_lmax=2,
                 ns=16, nv=4, num_conv_layers=2, lig_max_radius=5, rec_max_radius=30, cross_max_distance=250,
                 center_max_distance=30, distance_embed_dim=32, cross_distance_embed_dim=32, no_torsion=False,
                 scale_by_sigma=True, norm_by_sigma=True, use_second_order_repr=False, batch_norm=True,
                 dynamic_max_cross=False, dropout=0.0, smooth_edges=False, odd_parity=False,
                 separate_noise_schedule=False, lm_embedding_type=False, confidence_mode=False,
                 confidence_dropout=0, confidence_no_batchnorm=False,
                 asyncronous_noise_schedule=False, affinity_prediction=False, parallel=1,
                 parallel_aggregators="mean max min std", num_confidence_outputs=1, fixed_center_conv=False,
                 atom_max_neighbors=None,
                 no_aminoacid_identities=False, flexible_sidechains=False, include_miscellaneous_atoms=False,
                 use_old_atom_encoder=False):
        super().__init__()
        unsupported = {"sh_lmax != 1": sh_lmax != 1, "use_second_order_repr": use_second_order_repr,
                       "odd_parity": odd_parity,
                       "separate_noise_schedule": separate_noise_schedule,
                       "asyncronous_noise_schedule": asyncronous_noise_schedule,
                       "affinity_prediction": affinity_prediction, "parallel > 1": parallel != 1,
                       "include_miscellaneous_atoms": include_miscellaneous_atoms}
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError("MI355X score model: unsupported configuration: " + ", ".join(bad))
        assert (not no_aminoacid_identities) or (lm_embedding_type is None), "no language model emb without identities"
        if num_conv_layers < 1:
            raise NotImplementedError("num_conv_layers >= 1 required")
        self.t_to_sigma = t_to_sigma
        self.device = device
        self.timestep_emb_func = timestep_emb_func
        self.in_lig_edge_features = in_lig_edge_features
        self.sigma_embed_dim = sigma_embed_dim
        self.lig_max_radius, self.rec_max_radius = lig_max_radius, rec_max_radius
        self.cross_max_distance, self.dynamic_max_cross = cross_max_distance, dynamic_max_cross
        self.center_max_distance = center_max_distance
        self.distance_embed_dim, self.cross_distance_embed_dim = distance_embed_dim, cross_distance_embed_dim
        self.ns, self.nv = ns, nv
        self.scale_by_sigma, self.norm_by_sigma = scale_by_sigma, norm_by_sigma
        self.no_torsion = no_torsion
        self.num_conv_layers = num_conv_layers
        self.fixed_center_conv = fixed_center_conv
        self.atom_max_neighbors = atom_max_neighbors
        self.no_aminoacid_identities = no_aminoacid_identities
        self.flexible_sidechains = flexible_sidechains
        self.confidence_mode = bool(confidence_mode)
        self.smooth_edges = bool(smooth_edges)      # cosine edge weights (engine._front / _torsion_head)

        enc = OldAtomEncoder if use_old_atom_encoder else AtomEncoder
        sd, dd, cd = sigma_embed_dim, distance_embed_dim, cross_distance_embed_dim
        self.lig_node_embedding = enc(ns, LIG_FEATURE_DIMS, sd)
        self.lig_edge_embedding = _mlp(in_lig_edge_features + sd + dd, ns, ns, dropout)
        self.rec_node_embedding = enc(ns, REC_RESIDUE_FEATURE_DIMS, sd, lm_embedding_type=lm_embedding_type)
        self.rec_edge_embedding = _mlp(sd + dd, ns, ns, dropout)
        self.atom_node_embedding = enc(ns, REC_ATOM_FEATURE_DIMS, sd)
        self.atom_edge_embedding = _mlp(sd + dd, ns, ns, dropout)
        self.lr_edge_embedding = _mlp(sd + cd, ns, ns, dropout)
        self.ar_edge_embedding = _mlp(sd + dd, ns, ns, dropout)
        self.la_edge_embedding = _mlp(sd + cd, ns, ns, dropout)
        self.lig_distance_expansion = GaussianSmearing(0.0, lig_max_radius, dd)
        self.rec_distance_expansion = GaussianSmearing(0.0, rec_max_radius, dd)
        self.cross_distance_expansion = GaussianSmearing(0.0, cross_max_distance, cd)

        def out_blocks(m):  # [(mul, dim, is_0e)] in irreps order 0e,1o,1e,0o
            return [(mul, dim, s) for mul, dim, s in ((m[0], 1, True), (m[1], 3, False), (m[2], 3, False), (m[3], 1, False))
                    if mul > 0]

        convs = []
        self._layer_specs, self._layer_specs_g = [], []
        # Source-node factorisation (packing.faster_tp_spec(factorized=True)): used for a conv when its edge set has on
        # average at least `factorize_min_degree` edges per source node (then streaming one G[j] per node is cheaper
        # than the per-edge MFMA work it replaces).  0 disables it (every conv on the direct path).
        self.factorize_min_degree = 3.0
        self.prune_last_receptor_layer = True   # layer L-2 receptor-side convs only where the final layer reads them
        self.share_layer0 = True       # layer-0 receptor-side convs once per batch of identical receptors (forward)
        self.share_clean_layer1 = True  # layer-1 atom<-atom messages between atoms no ligand message has reached: once (forward)
        self.share_flex_layer0 = True   # flexible side chains: layer-0 atom-side convs per sample only where an atom moved nearby
        # ... when there is enough to save: its lists cost ~20 launches (atom-atom edges x ns; cfg1 x 4 samples = 0.57 M: 1.10 ms
        # per step without, 1.20 with; cfg2 x 5 samples = 2.7 M: 6.29 -> 6.14 ms; x 40: 39.4 -> 37.2 ms)
        self.flex_share_min_work = 2_000_000
        # The index lists of these eliminations are built on the device (engine._lists: ~25 small launches, no host
        # synchronisation); below this many atom-atom edges they are skipped (0: always on)
        self.plan_min_edges = 0
        self._static_cache = {}        # see _cached()
        self.before_layers = None      # optional callable, run once per forward between the front (graphs, edge embeddings,
                                       # CSR views) and the conv layers: sampler.PipelinedSampler orders the layers of its
                                       # resident groups with it (an event wait on the current stream)
        self.cache_slot = 0            # callers that alternate between several resident batches (sampler.PipelinedSampler)
                                       # give each its own slot so that they do not evict each other's entries
        self._stage_a_stacks = {}      # (layer, conv ids) -> stacked stage-A right-hand sides, see _stage_a()
        self.section_timer = None      # optional SectionTimer (tools/time_sections.py): per-section GPU + host time
        self.check_weight_values = True  # see _refresh_weight_caches
        self.range_check_in_forward = True   # forward(): read the h2 range flag before returning and recover in fp32 (samplers: per run)
        # Capacity of the ligand<-atom edge list: pocket atoms within lig_max_radius of a ligand atom, per ligand atom (its
        # worst case, every atom of the pocket, is ~80 x what occurs: the grids of every consumer are sized for the capacity).
        # 128 is ~3 x the densest packing of protein heavy atoms inside 5 A; a search that finds more sets a flag in pinned host
        # memory and the NEXT forward (or Sampler.run's end) raises - results are never silently truncated.
        self.la_capacity_per_atom = 128
        # Small batches (a strong-scaling shard: 5 of 40 samples per GPU): a layer's independent launches run side by side on
        # forked streams (engine._Fork); above this many pocket atoms in the batch every kernel fills the chip on its own
        self.concurrent_small_batches = True
        self.concurrent_max_atoms = 16000
        self.fork_small_means = True     # ... and a layer's three segmented means (ligand / atom / receptor rows) side by side
        # Large batches: the layer's chains as parallel branches of the captured step (engine._layers, "pipelined": the direct conv of
        # layer l + 1 starts as soon as the atom and receptor means of layer l are queued; receptor / ligand / atom chains
        # [mean -> stage A] side by side); same kernels, same arguments, same bits as the serial order (False)
        self.overlap_direct_conv = True
        # ... and the layer's factorised convs as TWO launches: the receptor- / ligand-sourced convs start as soon as the three means and their
        # own stage A are done, beside stage A of the atom rows (the layer's largest product); the atom-sourced convs follow behind it.
        # Same kernels, same tasks, same bits (a conv's workgroups do not depend on the launch it sits in); 17.5 -> 17.3 ms rigid,
        # 21.8 -> 21.4 ms with flexible side chains (profiles/r05_conv32_ab.txt)
        self.split_rows_launch = True
        self.shape_early_rows = False           # the early launch at ONE workgroup per CU beside stage A of the atom rows (measured: off)
        self.split_rows_min_g_bytes = 3.0e9     # ... where stage A of the atom rows writes at least this much (engine._layers): 4.7 GB at
        # 40 samples of cfg2 (3.6 GB in the 3-byte plane form); at 20 samples (2.4 GB; 1.8) the second launch cost 0.1 - 0.2 of 9.2 ms, on the
        # README's small model (1.4 GB) 0.3 of 3.2
        # The front's independent chains side by side (parallel branches of the captured step; same kernels, same arguments, same bits):
        # [node encoders -> edge embeddings] beside [neighbour searches -> CSR / source-ordered views], and - rigid receptor - the index
        # lists of the work eliminations (first read by layer 1) beside stage A + the 32-edge conv launch of layer 0 (engine._front,
        # engine._forward)
        self.fork_front = True
        self.fork_lists_flex = True     # flexible side chains: the pruned lists of the last layers beside the layer-0 / layer-1 lists
        self.concurrent_heads = True   # the torsion read-outs on forked streams beside the tr / rot read-out (any batch size)
        # Option, OFF: stage A on the bf16 matrix cores with both operands split into three bfloat16 terms (csrc/ddp_gemm.hip,
        # ddp_stage_a_x3_kernel): fp32-class accuracy (error <= 2^-21 sum |x w|, measured against fp64 in
        # tests/test_gpu_parity.py::test_stage_a_bf16x3_error) at 1/2.7 of the matrix time.  Measured on one box, alternating
        # (profiles/r03_stage_a_bf16x3_ab.txt): stage A 5.77 -> 5.54 ms per 40-sample step, but the 32-edge conv kernel that
        # follows every stage A runs 5 - 6 % slower (3.26 -> 3.45 ms per launch: the chip holds a lower clock after the bf16
        # bursts) and the step gets 1 ms LONGER (32.0 -> 33.0 ms); at 5 samples 5.48 -> 5.44 ms.  False = exact fp32 MFMA.
        self.stage_a_bf16x3 = False
        # Stage A on the fp16 matrix cores with both operands split in two halves (hi + lo / 2048, three products per 16 k: the conv
        # kernels' h2 form, csrc/ddp_gemm.hip ddp_stage_a_h2_kernel): 1/5 of the matrix time, 22-bit operands (error <= 2^-20 sum|x w|, the fp32 MFMA chain's class)
        self.stage_a_h2 = True
        self._overflow_flag = None
        self.exact_sizes = False       # test mode: device-side list sizes are read back and every list is cut to its length
        self.debug_conv_outputs = None  # set to a dict: forward then stores the output [n_out, d_out] of every conv call in it
                                        # (conv_layers.<9l+k>, final_conv, tor_bond_conv, sc_tor_bond_conv: the tensors the
                                        # reference's forward hooks see, tests/golden `conv_stats`) and runs the general path
                                        # (no layer-0 sharing, no clean-pair sharing, no dead-output pruning)
        for i in range(num_conv_layers):
            mi, mo = P.irreps_muls(ns, nv, i), P.irreps_muls(ns, nv, i + 1)
            spec = P.faster_tp_spec(mi, mo, 3 * ns)
            spec_g = P.faster_tp_spec(mi, mo, 3 * ns, factorized=True)
            self._layer_specs.append(spec)
            self._layer_specs_g.append(spec_g)
            for _ in range(9):
                convs.append(TensorProductConvLayer(spec, out_blocks(mo), batch_norm=batch_norm, dropout=dropout,
                                                    spec_g=spec_g))
        self.conv_layers = nn.ModuleList(convs)
        m_final = P.irreps_muls(ns, nv, num_conv_layers)
        self._d_final = P.irreps_dim(m_final)
        self._ldx = (P.irreps_dim(P.irreps_muls(ns, nv, num_conv_layers)) + 3) // 4 * 4

        if self.confidence_mode:
            # confidence head (reference models/all_atom_score_model.py:124-146); host-side PyTorch: a [B, <=4ns] MLP
            conf_in = (2 * ns if num_conv_layers >= 3 else ns) * (2 if flexible_sidechains else 1)
            bn = (lambda: nn.Identity()) if confidence_no_batchnorm else (lambda: nn.BatchNorm1d(ns))
            self.confidence_predictor = nn.Sequential(
                nn.Linear(conf_in, ns), bn(), nn.ReLU(), nn.Dropout(confidence_dropout),
                nn.Linear(ns, ns), bn(), nn.ReLU(), nn.Dropout(confidence_dropout),
                nn.Linear(ns, num_confidence_outputs))
        else:
            self.center_distance_expansion = GaussianSmearing(0.0, center_max_distance, dd)
            self.center_edge_embedding = _mlp(dd + sd, ns, ns, dropout)
            self.final_conv = TensorProductConvLayer(P.faster_tp_spec(m_final, (0, 2, 2, 0), 2 * ns),
                                                     [(2, 3, False), (2, 3, False)], batch_norm=batch_norm, dropout=dropout)
            self.tr_final_layer = nn.Sequential(nn.Linear(1 + sd, ns), nn.Dropout(dropout), nn.ReLU(), nn.Linear(ns, 1))
            self.rot_final_layer = nn.Sequential(nn.Linear(1 + sd, ns), nn.Dropout(dropout), nn.ReLU(), nn.Linear(ns, 1))
            tor_blocks = [(ns, 1, False), (ns, 1, True)]   # "ns x0o + ns x0e": 0o first
            if not no_torsion:
                self.final_edge_embedding = _mlp(dd, ns, ns, dropout)
                self.tor_bond_conv = TensorProductConvLayer(P.torsion_tp_spec(m_final, ns, 3 * ns), tor_blocks,
                                                            batch_norm=batch_norm, dropout=dropout)
                self.tor_final_layer = nn.Sequential(nn.Linear(2 * ns, ns, bias=False), nn.Tanh(), nn.Dropout(dropout),
                                                     nn.Linear(ns, 1, bias=False))
            if flexible_sidechains:
                self.sidechain_final_edge_embedding = _mlp(dd, ns, ns, dropout)
                self.sc_tor_bond_conv = TensorProductConvLayer(P.torsion_tp_spec(m_final, ns, 3 * ns), tor_blocks,
                                                               batch_norm=batch_norm, dropout=dropout)
                self.sc_tor_final_layer = nn.Sequential(nn.Linear(2 * ns, ns, bias=False), nn.Tanh(), nn.Dropout(dropout),
                                                        nn.Linear(ns, 1, bias=False))
        with np.load(os.path.join(ASSETS, "score_norm_tables.npz")) as z:
            self._so3_table = torch.from_numpy(z["so3_exp_score_norms"]).float()
            self._torus_table = torch.from_numpy(z["torus_score_norm"]).float()
        self._edge_packs: Dict[str, _EdgeMLPPack] = {}
        self.last_stats: Dict[str, float] = {}
        # plane form of the factorised convs' G (property g_planes3); DDP_G_PLANES3 = 0 / 1 in the environment sets the default of every model
        # built in the process (the parity suites under the other form: profiles/r06_g3byte_parity.txt)
        self.rows_mfma16 = os.environ.get("DDP_ROWS_MFMA16", ROWS_MFMA16_DEFAULT) == "1"
        self.direct_rows = os.environ.get("DDP_DIRECT_ROWS", DIRECT_ROWS_DEFAULT) == "1"
        # a direct conv through the rows kernel as up to this many tasks of segment ranges where its 128-edge workgroups would not fill the chip
        # (engine.direct_tasks: ceil(512 / workgroups), at most this; 1 = never split)
        self.direct_rows_max_split = int(os.environ.get("DDP_DIRECT_SPLIT", "6"))
        self.g_planes3 = os.environ.get("DDP_G_PLANES3", G_PLANES3_DEFAULT) == "1"

    # ---- checkpoint compatibility -------------------------------------------------------------
    _IGNORED_PREFIXES = ("final_tp_tor.", "final_tp_sc_tor.", "tor_bond_conv.tp.", "sc_tor_bond_conv.tp.")

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Reference checkpoints carry e3nn-internal buffers under final_tp_tor.* / *.tp.* (SURVEY §8(c)); they hold
        no learnable state and are dropped."""
        sd = {k: v for k, v in state_dict.items() if not k.startswith(self._IGNORED_PREFIXES)}
        out = super().load_state_dict(sd, strict=strict, **kw)
        self.invalidate_packed()
        return out

    def _shared_receptor_side(self, B, rec, atom, rpos, apos, lay_r, lay_a, rr, ar, aa, atoms=True):
        """Which receptor-side convs see the SAME problem in every graph of the batch (the usual sampling batch: N poses of
        one complex): per conv k in (3 atom<-atom, 5 atom<-receptor, 6 receptor<-receptor, 8 receptor<-atom) either None
        or (receiver nodes per graph, edges per graph, source nodes per graph).  Exact comparison of node features,
        positions and per-graph edge lists; depends only on step-independent tensors, so it is evaluated once (`_cached`).
        atoms=False: the atom side is not examined (flexible side chains move per sample and per step: its comparison
        would fail anyway, after a handful of host synchronisations on every call); atoms="static": only what does not move
        is examined (features, atom-receptor edges) and reported under the key "flex"."""
        out = {3: None, 5: None, 6: None, 8: None}
        if B < 2 or not (lay_r.uniform and lay_a.uniform):
            return out
        nr, na = lay_r.nmax, lay_a.nmax

        def same_rows(t, n):
            v = t.reshape(B, n, -1)
            return bool((v == v[:1]).all())

        def same_edges(ei, n0, n1):
            E = ei.shape[1]
            if E == 0 or E % B:
                return 0
            e = E // B
            off = torch.arange(B, device=ei.device).unsqueeze(1)
            a, b = ei[0].reshape(B, e) - off * n0, ei[1].reshape(B, e) - off * n1
            ok = (a == a[:1]).all() & (b == b[:1]).all() & (a >= 0).all() & (a < n0).all() & (b >= 0).all() & (b < n1).all()
            return e if bool(ok) else 0

        rec_same = same_rows(rec.x, nr) and same_rows(rpos, nr)
        atom_same = atoms is True and same_rows(atom.x, na) and same_rows(apos, na)
        if rec_same:
            e = same_edges(rr, nr, nr)
            out[6] = (nr, e, nr) if e else None
        if atoms == "static" and rec_same and same_rows(atom.x, na):
            # flexible side chains: the atoms' FEATURES and the atom-receptor edges repeat across the samples, their positions
            # (and with them the atom kNN graph) do not - what can still be shared is decided per step (engine._lists)
            e = same_edges(ar, na, nr)
            out["flex"] = (na, e, nr) if e else None
        if atom_same:
            e = same_edges(aa, na, na)
            out[3] = (na, e, na) if e else None
        if rec_same and atom_same:
            e = same_edges(ar, na, nr)
            if e:
                out[5], out[8] = (na, e, nr), (nr, e, na)
        return out

    def _weight_tensors(self):
        ts = self.__dict__.get("_weight_tensors_")
        if ts is None:   # (the list is rebuilt when modules change device / dtype: _apply)
            ts = self.__dict__["_weight_tensors_"] = list(self.parameters()) + list(self.buffers())
        return ts

    def _weights_version(self):
        """Sum of the autograd version counters of every parameter and buffer: in-place updates THROUGH the parameter
        (optimizer steps, `param.copy_` under no_grad, BatchNorm buffer edits) bump it.  Updates through `param.data` do NOT:
        `.data` is a detached alias with a version counter of its own - and that is what the reference's EMA does
        (utils/utils.py:216,239 `param.data.copy_`) - so the version sum is only the free first line; `_weights_fingerprint`
        is the second."""
        return sum(t._version for t in self._weight_tensors())

    def _weights_fingerprint(self):
        """Value fingerprint of every floating-point parameter and buffer: their L1 and L2 norms (two multi-tensor launches,
        one device-to-host copy of ~2 x 600 numbers = one host synchronisation).  Catches what the version counters cannot
        see: `param.data.copy_`, `ema.copy_to(model.parameters())`, `ema.restore(...)` (reference utils/utils.py:206-240)."""
        ts = [t.detach() for t in self._weight_tensors() if t.is_floating_point() and t.numel() > 0]
        if not ts:
            return ()
        parts = torch._foreach_norm(ts, 1) + torch._foreach_norm(ts, 2)
        return tuple(torch.stack([p.double() for p in parts]).tolist())

    def _refresh_weight_caches(self):
        """Packed weights, edge-MLP packs, stage-A stacks and the cached encoder parts bake the weights in: dropped when any
        parameter / buffer changed since they were built.  Called at the top of every forward; `check_weight_values = False`
        (set by sampler.Sampler around the steps of one run, after one full check) skips the value fingerprint and its host
        synchronisation."""
        wv = self._weights_version()
        fp = self._weights_fingerprint() if self.check_weight_values else self.__dict__.get("_weights_seen_fp")
        if self.__dict__.get("_weights_seen") != wv or self.__dict__.get("_weights_seen_fp") != fp:
            self.invalidate_packed()
            self._weights_seen = wv
            self._weights_seen_fp = fp if self.check_weight_values else self._weights_fingerprint()

    def overflow_flag(self, dev):
        """int32 [1] in pinned host memory (device-writable, host-readable without a synchronisation)."""
        if self._overflow_flag is None:     # [0]: ligand<-atom list truncated, [1]: a value outside the fp16 range met the h2 kernels
            self._overflow_flag = torch.zeros(2, dtype=torch.int32).pin_memory()
        return self._overflow_flag

    def check_overflow(self, range_too=True):
        """Raises for what the flag block (pinned host memory, written by the kernels) holds from forwards that have run: a truncated
        ligand<-atom list and / or a value outside the fp16 range in the h2 kernels of a forward whose result was NOT recovered (forwards
        entered through `forward` and samplers' runs recover by themselves: `_forward_recovering`, `Sampler.run`).  The whole block is
        cleared first, so a later, healthy forward is not blamed."""
        if self._overflow_flag is None:
            return
        # (range_too = False: the caller - a sampler's step - checks and recovers the range flag itself, once per run: it is left as it is)
        trunc, rng = int(self._overflow_flag[0]) != 0, range_too and int(self._overflow_flag[1]) != 0
        if not (trunc or rng):
            return
        if range_too:
            self._overflow_flag.zero_()
        else:
            self._overflow_flag[0] = 0
        msgs = []
        if rng:
            msgs.append("a node feature / edge feature / fc activation / G value outside the fp16 range (|v| > 65504, or NaN) reached the fp16 "
                        "hi/lo split kernels of an earlier forward whose scores were consumed without the range check (a hand-driven "
                        "Sampler.step sequence without Sampler.check_overflow): they are not to be trusted - model.conv_h2 = False runs the "
                        "exact fp32 MFMA form")
        if trunc:
            msgs.append(f"a ligand atom had more than la_capacity_per_atom = {self.la_capacity_per_atom} pocket atoms within lig_max_radius: "
                        f"the ligand<-atom edge list of an earlier forward was truncated; raise the capacity")
        raise L.DdpError("; ".join(msgs))

    def range_flag_raised(self, clear=True):
        """Did an h2 kernel of a forward that has RUN (the caller synchronised) meet a value it cannot split?"""
        if self._overflow_flag is None or int(self._overflow_flag[1]) == 0:
            return False
        if clear:
            self._overflow_flag[1] = 0
        return True

    @property
    def conv_h2(self):
        """The fc products (and stage A) as fp16 hi/lo split products on the fp16 matrix cores (22-bit operands, fp32 accumulation;
        csrc/ddp_conv.hip, ddp_conv_rows.hip); False: the exact fp32 MFMA kernels of rounds 1 - 3.  The split form is not total - a value
        beyond +-65504 cannot be split - so a forward that meets one is run again in the fp32 form before its result is returned
        (`forward`, `Sampler.run`); this switch only chooses what is TRIED first.  Changing it drops captured steps (they carry the
        kernels' form)."""
        return self.__dict__.get("_conv_h2", True)

    @conv_h2.setter
    def conv_h2(self, value):
        value = bool(value)
        if value != self.conv_h2:
            self.__dict__["_conv_h2"] = value
            self.__dict__["_packed_epoch"] = self.__dict__.get("_packed_epoch", 0) + 1

    @property
    def g_planes3(self):
        """Plane form of the factorised convs' G (ddp_conv_task_t::gh_fmt).  False: fp16 hi + fp16 lo words, 4 bytes per value, 22 significant
        bits.  True: fp16 hi (truncated) + a continuation byte (ddp_stage_a_gh3), 3 bytes per value, 19 significant bits: a quarter less of
        the step's G round trip through HBM (written by stage A, read once by ddp_conv_rows), and stage A leaves whole 128-byte lines.
        Precision: 3.5e-6 worst on the scores of the golden cases, where the 22-bit planes give 5.5e-6 (the floor the rest of the path
        sets; profiles/r06_g19bit_precision.txt).  Changing it drops the packed weights and captured steps.  (ABI 16 had an e4m3 byte in
        this place: 15 - 16 bits, 2.4e-4 - outside the bar; profiles/r06_g3byte_parity.txt.)"""
        return bool(self.__dict__.get("_g_planes3", False))

    @g_planes3.setter
    def g_planes3(self, value):
        value = bool(value)
        if value != self.g_planes3:
            self.__dict__["_g_planes3"] = value
            self._stage_a_stacks = {}
            for m_ in self.modules():
                if isinstance(m_, TensorProductConvLayer):
                    m_.gh_fmt = 1 if value else 0
                    m_._packed_g = None
            self.__dict__["_rows_checked_epoch"] = None
            self.__dict__["_packed_epoch"] = self.__dict__.get("_packed_epoch", 0) + 1

    @property
    def rows_mfma16(self):
        """Operand images of the row-stationary conv kernel (ddp_conv_task_t::rows_form).  True: every tile product on
        v_mfma_f32_16x16x32_f16 (csrc/ddp_conv_rows16.hip); False: v_mfma_f32_32x32x16_f16 (csrc/ddp_conv_rows.hip).  Same arithmetic per
        product (unified fp16 hi/lo planes, fp32 accumulation), other summation order inside a k-step: results agree to fp32 rounding, not
        bit for bit.  Changing it drops the packed weights and captured steps."""
        return bool(self.__dict__.get("_rows_mfma16", False))

    @rows_mfma16.setter
    def rows_mfma16(self, value):
        value = bool(value)
        if value != self.rows_mfma16:
            self.__dict__["_rows_mfma16"] = value
            self._stage_a_stacks = {}
            for m_ in self.modules():
                if isinstance(m_, TensorProductConvLayer):
                    m_.rows_form = 1 if value else 0
                    m_._packed_g = None
                    m_._packed = None
            self.__dict__["_rows_checked_epoch"] = None
            self.__dict__["_packed_epoch"] = self.__dict__.get("_packed_epoch", 0) + 1

    @property
    def direct_rows(self):
        """The layers' DIRECT convs (receptor<-atom: one edge per atom, nothing to factorise) through the row-stationary kernel as well
        (csrc/ddp_conv_rows16.hip: every feature a stream tile, the fc.3 bias in the padding k row - ddp_conv_task_t::rows_bias_k), where
        the shape allows it (rows_mfma16 on, hid = 180); False: ddp_conv_messages (64-edge workgroups).  Changing it drops the packed weights
        and captured steps."""
        return bool(self.__dict__.get("_direct_rows", False))

    @direct_rows.setter
    def direct_rows(self, value):
        value = bool(value)
        if value != self.direct_rows:
            self.__dict__["_direct_rows"] = value
            for m_ in self.modules():
                if isinstance(m_, TensorProductConvLayer):
                    m_.direct_rows = value
                    m_._packed_g = None
                    m_._packed = None
            self.__dict__["_packed_epoch"] = self.__dict__.get("_packed_epoch", 0) + 1

    def rows_all_or_none(self, device):
        """ddp_conv_rows runs either every factorised conv of its size class or none: stage A writes the G of several convs in one
        launch, in ONE layout.  A conv whose weights the kernel's operand planes cannot hold (|w| > 255, packing.rows_stream) therefore
        switches the model's factorised convs back to the 32-edge kernel.  Checked once per set of packed weights."""
        if self.__dict__.get("_rows_checked_epoch") == self.__dict__.get("_packed_epoch", 0):
            return
        convs = [m for m in self.modules() if isinstance(m, TensorProductConvLayer) and getattr(m, "spec_g", None) is not None
                 and P.rows_supported(m.spec_g)]
        pks = [c.packed_g(device) for c in convs]
        if any(pk.wsh is None for pk in pks):
            for pk in pks:
                pk.wsh = pk.bsp = pk.wgh = pk.gh_groups = pk.gh_ld = None
        self.__dict__["_rows_checked_epoch"] = self.__dict__.get("_packed_epoch", 0)

    def invalidate_packed(self):
        # (a captured step holds the ADDRESSES of what is dropped here: sampler.Sampler compares this counter before a replay)
        self.__dict__["_packed_epoch"] = self.__dict__.get("_packed_epoch", 0) + 1
        self.__dict__["_rows_checked_epoch"] = None
        self._weights_seen = None
        self._weights_seen_fp = None
        self.__dict__["_weight_tensors_"] = None
        self._stage_a_stacks = {}
        for m in self.modules():
            if isinstance(m, TensorProductConvLayer):
                m._packed = None
                m._packed_g = None
        self._edge_packs = {}
        self._static_cache = {}

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self.invalidate_packed()
        return out

    @property
    def _static_cache(self):
        """The `_cached` entries of the current `cache_slot`."""
        return self.__dict__.setdefault("_static_caches", {}).setdefault(self.__dict__.get("cache_slot", 0), {})

    @_static_cache.setter
    def _static_cache(self, value):   # assigning {} drops the entries of every slot
        self.__dict__["_static_caches"] = {self.__dict__.get("cache_slot", 0): value} if value else {}

    # ---- small host-side pieces ---------------------------------------------------------------
    def _edge_pack(self, name, rbf_slice, dev) -> _EdgeMLPPack:
        pk = self._edge_packs.get(name)
        if pk is None or pk.w1d.device != dev:
            pk = self._edge_packs[name] = _EdgeMLPPack(getattr(self, name), rbf_slice, dev)
        return pk

    def _sigma_spec(self, t, dev):
        """How ddp_node_linear gets the sigma embedding of the times `t`: evaluated in the kernel when timestep_emb_func is
        this package's sinusoidal embedding (diffusion.get_timestep_embedding), otherwise computed by calling it."""
        from .diffusion import _frequencies, sinusoidal_embedding
        f = self.timestep_emb_func
        kw = getattr(f, "keywords", None) or {}
        if getattr(f, "func", None) is sinusoidal_embedding and not getattr(f, "args", ()) and "dim" in kw \
                and set(kw) <= {"dim", "scale", "max_positions"} and kw["dim"] >= 4:
            return ("t", t.float(), float(kw.get("scale", 1.0)), _frequencies(kw["dim"] // 2, kw.get("max_positions", 10000), dev),
                    int(kw["dim"]))
        return ("emb", f(t).float().contiguous())

    def _node_tables(self, lig, rec, atom, dev):
        """Node encoders (all_atom_score_model.py:249,254,259 -> models/score_model.py:54-82 / :17-52), the sinusoidal sigma
        embedding of every node (:453,495,520) and the per-node `pre` tables of the edge-embedding MLPs (the part of their
        first Linear that depends on the node only: W1[:, sigma columns] @ node_sigma_emb + b1) in ONE ddp_node_linear launch
        (two when an OldAtomEncoder carries an ESM block).  Returns the node-feature arrays [N, ldx] (encoder output in the
        first ns columns, zeros behind) and {edge set: [N, ns] view of its pre table}."""
        ns, ldx, sd = self.ns, self._ldx, self.sigma_embed_dim
        nf, dd = self.in_lig_edge_features, self.distance_embed_dim
        pre_specs = {"ligand": [("ll", "lig_edge_embedding", nf), ("lr", "lr_edge_embedding", 0), ("la", "la_edge_embedding", 0)]
                     + ([] if self.confidence_mode else [("center", "center_edge_embedding", dd)]),
                     "receptor": [("rr", "rec_edge_embedding", 0)],
                     "atom": [("aa", "atom_edge_embedding", 0), ("ar", "ar_edge_embedding", 0)]}
        jobs, jobs2, xs, pre, keep = [], [], [], {}, []
        for name, st, enc in (("ligand", lig, self.lig_node_embedding), ("receptor", rec, self.rec_node_embedding),
                              ("atom", atom, self.atom_node_embedding)):
            pk = self._edge_packs.get("enc_" + name)
            if pk is None or pk.table.device != dev:
                pk = self._edge_packs["enc_" + name] = _EncoderPack(enc, dev)
            N = st.x.shape[0]
            ncat, n_lm = pk.n_cat, st.x.shape[1] - pk.n_cat
            cat = self._cached("cat_" + name, (st.x,), lambda st=st, ncat=ncat: st.x[:, :ncat].to(torch.int32).contiguous())
            xf = None
            if n_lm > 0:
                xf = self._cached("xf_" + name, (st.x,), lambda st=st: st.x.float().contiguous())
            sigma = self._sigma_spec(st.node_t["tr"], dev)
            sig_out = torch.empty((N, sd), device=dev)
            st.node_sigma_emb = sig_out                       # (:453,495,520) the reference leaves it on the batch
            x = torch.empty((N, ldx), device=dev)
            xs.append(x)
            keep += [cat, xf, sigma]
            if not pk.old:
                if not pk.has_extra:
                    raise NotImplementedError("AtomEncoder without additional features (sigma_embed_dim = 0, no ESM)")
                if pk.w.shape[0] != pk.emb_dim + n_lm + sd:
                    raise ValueError(f"{name}.x has {st.x.shape[1]} columns, the encoder expects {pk.w.shape[0] - pk.emb_dim - sd + ncat}")
                # out = [emb | ESM | sigma] @ W + b: everything but the sigma columns is the same at every denoising step of a
                # batch (the ESM block is 1280 of the receptor's 1340 input columns) - computed once per batch by a job of its
                # own, kept while `x` and the weights are unchanged, and ADDED by the per-step job (K = sigma_embed_dim)
                k_st = pk.emb_dim + n_lm

                def static_part(N=N, pk=pk, cat=cat, xf=xf, ncat=ncat, n_lm=n_lm):
                    out = torch.empty((N, ns), device=dev)
                    _launch_node_jobs([_node_job(N, out, ns, pk.w, pk.b, cat=cat, pack=pk, emb_mode=1,
                                                 dense=[(xf, ncat, n_lm)] if n_lm else [])])
                    return out
                static = self._cached("enc_static_" + name, (st.x,), static_part)
                keep.append(static)
                jobs.append(_node_job(N, x, ns, pk.w[k_st:], None, zero_to=ldx, sigma=sigma, sig_out=sig_out, add=static))
            elif not pk.has_lm:
                jobs.append(_node_job(N, x, ns, pk.w1, pk.b1, zero_to=ldx, cat=cat, pack=pk, emb_mode=2, sigma=sigma, sig_out=sig_out))
            else:   # scalars = the first n_scalar columns behind the categorical ones, "ESM" = the last 1280 of [ESM | sigma]
                tmp = torch.empty((N, ns), device=dev)
                keep.append(tmp)
                jobs.append(_node_job(N, tmp, ns, pk.w1, pk.b1, cat=cat, pack=pk, emb_mode=2, dense=[(xf, ncat, pk.n_scalar)]))
                jobs2.append(_node_job(N, x, ns, pk.w2, pk.b2, zero_to=ldx,
                                       dense=[(tmp, 0, ns), (xf, ncat + n_lm + sd - 1280, 1280 - sd)], sigma=sigma, sig_out=sig_out))
            specs = pre_specs[name]
            ppk = self._edge_packs.get("pre_" + name)
            if ppk is None or ppk[0].device != dev:
                ws = [getattr(self, mlp)[0].weight.detach().float()[:, s0:s0 + sd].t() for _, mlp, s0 in specs]
                bs = [getattr(self, mlp)[0].bias.detach().float() for _, mlp, _ in specs]
                ppk = self._edge_packs["pre_" + name] = (torch.cat(ws, 1).contiguous().to(dev), torch.cat(bs).contiguous().to(dev))
            buf = torch.empty((N, len(specs) * ns), device=dev)
            jobs.append(_node_job(N, buf, len(specs) * ns, ppk[0], ppk[1], sigma=sigma))
            for i, (key, _, _) in enumerate(specs):
                pre[key] = buf[:, i * ns:(i + 1) * ns]
        _launch_node_jobs(jobs)
        if jobs2:
            _launch_node_jobs(jobs2)
        self._keep_alive = keep      # raw pointers were handed to the launches above
        return xs[0], xs[1], xs[2], pre

    def _sigma_ranges(self):
        """((sigma_min, sigma_max) x 4) when `t_to_sigma` is this package's schedule bound to its ranges (the reference's
        functools.partial(t_to_sigma_compl, args=args), utils/utils.py / inference.py) - ddp_step_prologue then evaluates it; any
        other callable is called as it is (None)."""
        from .diffusion import t_to_sigma
        f = self.t_to_sigma
        kw = getattr(f, "keywords", None) or {}
        if getattr(f, "func", None) is not t_to_sigma or getattr(f, "args", ()) or set(kw) != {"args"}:
            return None
        r = kw["args"]
        try:
            out = tuple((float(getattr(r, p + "_sigma_min")), float(getattr(r, p + "_sigma_max"))) for p in ("tr", "rot", "tor", "sidechain_tor"))
        except (AttributeError, TypeError, ValueError):
            return None
        return out if all(lo > 0 and hi > 0 for lo, hi in out) else None

    def _head_weights(self, dev):
        """fp32 device tensors of the read-out MLPs and score-norm tables as ddp_trrot_head / ddp_tor_head take them (kept with
        the packed weights: rebuilt when a parameter changes)."""
        hw = self._edge_packs.get("heads")
        if hw is None or hw["dev"] != dev:
            f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()      # noqa: E731
            hw = {"dev": dev}
            for name in ("tr_final_layer", "rot_final_layer"):
                seq = getattr(self, name)
                hw[name] = (f(seq[0].weight), f(seq[0].bias), f(seq[3].weight.reshape(-1)), f(seq[3].bias))
            for name in ("tor_final_layer", "sc_tor_final_layer"):
                seq = getattr(self, name, None)
                if seq is not None:
                    hw[name] = (f(seq[0].weight), f(seq[3].weight.reshape(-1)))
            # utils/so3.py:85-89 and utils/torus.py:78-82: index arithmetic in float32 like numpy on a float32 array
            hw["so3"], hw["torus"] = f(self._so3_table), f(self._torus_table)
            lo, hi = math.log10(0.01), math.log10(2.0)
            hw["so3_lo"], hw["so3_span"] = float(np.float32(lo)), float(np.float32(hi - lo))
            lo, hi = math.log(3e-3), math.log(2.0)
            hw["torus_lo"], hw["torus_span"] = float(np.float32(lo)), float(np.float32(hi - lo))
            self._edge_packs["heads"] = hw
        return hw

    def _so3_score_norm(self, sigma):
        """reference utils/so3.py:85-89 (float32 arithmetic like numpy on a float32 array)."""
        lo, hi, n = math.log10(0.01), math.log10(2.0), 1000
        idx = (torch.log10(sigma.float()) - np.float32(lo)) / np.float32(hi - lo) * n
        idx = torch.clamp(torch.round(idx).long(), 0, n - 1)
        if self._so3_table.device != sigma.device:
            self._so3_table = self._so3_table.to(sigma.device)
        return self._so3_table[idx]

    def _torus_score_norm(self, sigma):
        """reference utils/torus.py:78-82."""
        lo, hi, n = math.log(3e-3), math.log(2.0), 5000
        s = torch.log(sigma.float() / np.float32(np.pi))
        s = (s - np.float32(lo)) / np.float32(hi - lo) * n
        idx = torch.round(torch.clamp(s, 0, n)).long()
        if self._torus_table.device != sigma.device:
            self._torus_table = self._torus_table.to(sigma.device)
        return self._torus_table[idx]

    def _cached(self, name, inputs, fn):
        """Results that depend only on `inputs` (tensors) are kept across forward calls while those tensors are unchanged:
        the receptor side of the graph (atom kNN graph, receptor / atom-receptor CSR views) is the same at every denoising
        step unless side chains move.  "Unchanged" = same storage address, shape, strides and dtype AND the same autograd
        version counter (every in-place torch op bumps it; views share it); the entry holds a reference to the tensors, so
        their storage cannot be recycled for something else while the entry is alive.  The reference recomputes these every
        call (models/all_atom_score_model.py:524,545-564); the values are identical."""
        key = tuple((t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype) for t in inputs)
        ver = tuple(t._version for t in inputs)
        hit = self._static_cache.get(name)
        if hit is not None and hit[0] == key and hit[1] == ver:
            return hit[3]
        val = fn()
        self._static_cache[name] = (key, ver, tuple(inputs), val)
        return val

    # ---- forward --------------------------------------------------------------------------------
    def forward(self, data):
        """reference models/all_atom_score_model.py:238-436: (tr_pred [B,3], rot_pred [B,3], tor_pred [sum T], sc_tor_pred [sum S]),
        or the confidence logits in confidence_mode.  The work is in diffdock_pocket_amd/engine.py."""
        eng = self.__dict__.get("_engine")
        if eng is None:
            from .engine import ForwardEngine
            eng = self.__dict__["_engine"] = ForwardEngine(self)
        out = eng.forward(data)
        if not (self.conv_h2 and self.range_check_in_forward) or torch.cuda.is_current_stream_capturing():
            return out
        # The fp16 hi/lo form is not total: the kernels report a value they cannot split (|v| > 65504) through a flag in pinned host
        # memory.  The reference's caller synchronises on the next line anyway (utils/sampling.py:122-125), so the flag is read HERE, in
        # the same call, and a raised one reruns this forward in the exact fp32 MFMA form - the caller never sees the spoiled scores.
        # (Samplers switch this per-forward check off around their steps and recover a whole run: Sampler.run.)
        torch.cuda.current_stream(data["ligand"].pos.device).synchronize()
        if not self.range_flag_raised():
            return out
        self.__dict__["h2_recoveries"] = self.__dict__.get("h2_recoveries", 0) + 1
        return self.forward_fp32(data)

    def split_form_error(self, data):
        """How far the fp16 hi/lo split form of the fc products (what `forward` runs) is from the exact fp32 MFMA form ON THIS BATCH AND
        THESE WEIGHTS: {output name: max |split - fp32| / max |fp32|}.  The split operands carry 22 significant bits only while their lo
        halves are normal fp16 numbers (include/ddp_hip.h, DDP_ROWS_S*: absolute floors below |V| = 0.125); the parity suites run on
        seeded random weights with O(1) BatchNorm-ed activations, so a user with a trained checkpoint whose magnitudes differ can ask here
        instead of trusting them (two forwards; typical values 1e-6 .. 1e-5, the path's tolerance is 1e-4).  A batch the split form cannot
        represent at all (range flag) reports {"recovered": True}: `forward` then returns the fp32 form by itself."""
        before = self.__dict__.get("h2_recoveries", 0)
        a = self.forward(data)
        if self.__dict__.get("h2_recoveries", 0) != before:
            return {"recovered": True}
        b = self.forward_fp32(data)
        if not isinstance(a, (tuple, list)):
            a, b = (a,), (b,)
        names = ("tr", "rot", "tor", "sc_tor") if len(a) == 4 else ("confidence",)
        out = {}
        for n_, x, y in zip(names, a, b):
            if y.numel():
                out[n_] = float((x.double() - y.double()).abs().max() / y.double().abs().max().clamp(min=1e-30))
        return out

    def forward_fp32(self, data):
        """This forward with every fc product in the exact fp32 MFMA form (launch.CONV_H2 off for the tasks it builds), whatever conv_h2
        says; the packed fp32 weights always exist, nothing is invalidated."""
        prev, prev_chk = self.__dict__.get("_conv_h2", True), self.range_check_in_forward
        self.__dict__["_conv_h2"], self.range_check_in_forward = False, False
        try:
            return self.forward(data)
        finally:
            self.__dict__["_conv_h2"], self.range_check_in_forward = prev, prev_chk
