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
Configurations outside the README models (sh_lmax != 1, second-order irreps, smooth_edges, odd_parity, separate or
asynchronous noise schedules, affinity prediction, parallel > 1) raise NotImplementedError instead of silently differing.
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
    __slots__ = ("w1p", "b1p", "w2p", "b2p", "bn_scale", "bn_shift", "wg", "bg", "g_in_off")


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

    def packed_g(self, device) -> _PackedConv:
        """Weights for the factorised path: fc.3 tiles of the vector-input features only + the GEMM right-hand sides
        that turn source-node scalars into G / Gb (packing.factor_weights)."""
        if self._packed_g is None or self._packed_g.w1p.device != device:
            base = self.packed(device)
            pk = _PackedConv()
            pk.w1p, pk.b1p, pk.bn_scale, pk.bn_shift = base.w1p, base.b1p, base.bn_scale, base.bn_shift
            w2p, b2p = P.pack_fc2(self.spec_g, self.fc[3].weight, self.fc[3].bias)
            if w2p.numel() == 0:
                w2p, b2p = torch.zeros(64), torch.zeros(32)
            pk.w2p, pk.b2p = w2p.to(device), b2p.to(device)
            wg, bg, offs = P.factor_weights(self.spec_g, self.fc[3].weight, self.fc[3].bias)
            pk.wg = [w.to(device) if w is not None else None for w in wg]
            pk.bg = [b.to(device) if b is not None else None for b in bg]
            pk.g_in_off = offs
            self._packed_g = pk
        return self._packed_g

    def node_tensors(self, pk: _PackedConv, x_src: torch.Tensor):
        """Stage A of the factorised conv: per-source-node rows [G | Gb | pad] = x_scalar @ Wg (ddp_stage_a)."""
        lib = L.load()
        g = [None, None]
        N = x_src.shape[0]
        for slot in (0, 1):
            if pk.wg[slot] is None:
                continue
            w = pk.wg[slot]
            g[slot] = torch.empty((N, w.shape[1]), device=x_src.device, dtype=torch.float32)
            offs = (C.c_int32 * 1)(pk.g_in_off[slot])
            L.check(lib.ddp_stage_a(x_src.data_ptr(), x_src.shape[1], N, offs, 1, w.data_ptr(), w.shape[0], w.shape[1],
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
            self._packed = pk
        return self._packed

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
            g = self.node_tensors(pk, x)
            task = _make_task(pk, x, x.shape[1], so, sh, [(ea, so.eid, ea.shape[1], ea.shape[1])], msg, g=g, pos=so.pos)
            _launch_convs(self.spec_g, [task], flops_spec=self.spec)
        else:
            task = _make_task(self.packed(dev), x, x.shape[1], csr, sh, [(ea, csr.eid, ea.shape[1], ea.shape[1])], msg)
            _launch_convs(self.spec, [task])
        out = torch.zeros((n_out, self.spec.d_out), device=dev, dtype=torch.float32)
        _launch_reduce(out, self.spec.d_out, n_out, self.spec.d_out, [(msg, csr, self.packed(dev))], accumulate=False)
        return out


# ------------------------------------------------------------------------------------------------ launch helpers
_DEVICE_INDEX = [0]      # set by _require_hip (the device of the batch): `torch.cuda.current_stream()` costs ~9 us of Python per call
                         # and a step makes ~90 of them; the raw-handle query in _stream() is a single C call


def _require_hip(t: torch.Tensor):
    if not t.is_cuda:
        raise L.DdpError("the MI355X score model runs on a HIP device only (no CPU/eager fallback); "
                         "move the batch to cuda:<n>")
    L.load()
    _DEVICE_INDEX[0] = t.device.index if t.device.index is not None else torch.cuda.current_device()


def _stream():
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(_DEVICE_INDEX[0]))


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _make_task(pk: _PackedConv, x_src, ldx_src, csr, sh, segs, msg, g=None, pos=None) -> L.ConvTask:
    """segs: [(tensor, idx_int32[E], ld, ncols)], concatenated into edge_attr_ in this order."""
    t = L.ConvTask()
    t.x_src, t.ldx_src, t.n_edges = x_src.data_ptr(), ldx_src, csr.n_edges
    t.src, t.eid, t.sh = csr.src.data_ptr(), csr.eid.data_ptr(), sh.data_ptr()
    for k in range(L.DDP_MAX_SEGS):
        if k < len(segs):
            ten, idx, ld, n = segs[k]
            t.seg_ptr[k], t.seg_idx[k], t.seg_ld[k], t.seg_n[k] = ten.data_ptr(), idx.data_ptr(), ld, n
        else:
            t.seg_ptr[k], t.seg_idx[k], t.seg_ld[k], t.seg_n[k] = 0, 0, 0, 0
    t.w1p, t.b1p, t.w2p, t.b2p = pk.w1p.data_ptr(), pk.b1p.data_ptr(), pk.w2p.data_ptr(), pk.b2p.data_ptr()
    t.msg = msg.data_ptr()
    for k in range(2):
        t.g[k] = g[k].data_ptr() if (g is not None and g[k] is not None) else 0
    t.pos = pos.data_ptr() if pos is not None else 0
    return t


class ConvProfiler:
    """Times every ddp_conv_messages launch with HIP events on the launch stream and tallies its algorithmic FLOPs
    (BASELINE.md §3 formula x the launch's actual edge count).  Used by bench.py for the roofline entry."""

    def __init__(self):
        self.events, self.flops, self.executed, self.kernel = [], [], [], []
        self.useful, self.edges, self.boundary = [], [], []   # per launch: useful FLOPs, edges, algorithmic boundary bytes
        self.hbm = {}   # HBM-bound kernels: name -> [(event0, event1, algorithmic bytes of the launch)]
        self.hbm_on = False   # their ~45 extra event pairs per step cost wall time: bench.py times them in extra steps

    def hbm_summary(self, name):
        """(launches, algorithmic bytes, ms) of an HBM-bound kernel (ddp_stage_a_mfma_kernel, ddp_segment_reduce_kernel)."""
        rec = self.hbm.get(name, [])
        return len(rec), float(sum(r[2] for r in rec)), float(sum(r[0].elapsed_time(r[1]) for r in rec))

    def summary(self, kernel=None):
        """(launches, algorithmic FLOPs, ms) over all launches or over those of one kernel instantiation
        ("ddp_conv32_kernel": factorised shapes, "ddp_conv_messages_kernel": direct shapes)."""
        sel = [i for i, k in enumerate(self.kernel) if kernel is None or k == kernel]
        ms = sum(self.events[i][0].elapsed_time(self.events[i][1]) for i in sel)
        return len(sel), float(sum(self.flops[i] for i in sel)), float(ms)

    def executed_flops(self, kernel=None):
        """FLOPs of the padded MFMA tiles + the G pass of factorised convs (a model of what is issued; the PMC pass counts it)."""
        return float(sum(e for e, k in zip(self.executed, self.kernel) if kernel is None or k == kernel))

    def useful_flops(self, kernel=None):
        """Useful fp32 FLOPs of the executed formulation without padding (packing.ConvSpec.useful_flops_per_edge)."""
        return float(sum(e for e, k in zip(self.useful, self.kernel) if kernel is None or k == kernel))

    def boundary_bytes(self):
        """Algorithmic bytes at the module boundary of the recorded conv calls (SURVEY section 8(d):
        4 (N_in D_in + E F + 4 E + N_out D_out) + 16 E per TensorProductConvLayer.forward call)."""
        return float(sum(self.boundary))


_PROFILER: Optional[ConvProfiler] = None


class SectionTimer:
    """Diagnostic: `model.section_timer = SectionTimer()` records a device event and the host clock at each section
    boundary of forward; `summary()` gives per-section (gpu_ms, host_ms) summed over the recorded calls."""

    def __init__(self):
        self.marks = []

    def mark(self, name):
        import time
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.marks.append((name, ev, time.perf_counter()))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for (n0, e0, t0), (n1, e1, t1) in zip(self.marks[:-1], self.marks[1:]):
            if n1 == "start":
                continue
            g, h = out.get(n1, (0.0, 0.0))
            out[n1] = (g + e0.elapsed_time(e1), h + (t1 - t0) * 1e3)
        return out


def set_conv_profiler(p: Optional[ConvProfiler]):
    global _PROFILER
    _PROFILER = p


def _launch_convs(spec: P.ConvSpec, tasks: List[L.ConvTask], flops_spec: Optional[P.ConvSpec] = None, node_bytes: float = 0.0):
    """node_bytes: 4 (N_in D_in + N_out D_out) summed over the launch's conv calls (only used by the profiler)."""
    lib = L.load()
    if not tasks:
        return
    arr = (L.ConvTask * len(tasks))(*tasks)
    shape = spec.ctypes_shape()
    prof = _PROFILER
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(lib.ddp_conv_messages(C.byref(shape), arr, len(tasks), _stream()), "ddp_conv_messages")
    if prof is not None:
        e1.record()
        prof.events.append((e0, e1))
        ne = sum(t.n_edges for t in tasks)
        prof.flops.append((flops_spec or spec).flops_per_edge() * ne)
        prof.executed.append((spec.mfma_flops_per_edge_executed() + 2 * spec.hid * sum(spec.g_cols)) * ne)
        prof.useful.append(spec.useful_flops_per_edge() * ne)
        prof.edges.append(ne)
        prof.boundary.append(ne * (4.0 * (flops_spec or spec).f_in + 32.0) + node_bytes)
        prof.kernel.append("ddp_conv32_kernel" if spec.factorized else "ddp_conv_messages_kernel")


def _launch_reduce(x, ldx, n_nodes, d_out, sources, accumulate=True):
    """sources: [(msg, csr, packed[, rowmap])] in the reference's summation order; rowmap (int32 per CSR position, optional)
    = the row of `msg` that holds the position's message."""
    lib = L.load()
    arr = (L.ReduceSrc * max(len(sources), 1))()
    for i, src_ in enumerate(sources):
        msg, csr, pk = src_[:3]
        arr[i].msg, arr[i].rowptr = msg.data_ptr(), csr.rowptr.data_ptr()
        arr[i].bn_scale, arr[i].bn_shift, arr[i].n_edges = pk.bn_scale.data_ptr(), pk.bn_shift.data_ptr(), csr.n_edges
        arr[i].rowmap = src_[3].data_ptr() if (len(src_) > 3 and src_[3] is not None) else 0
    prof = _PROFILER if (_PROFILER is not None and _PROFILER.hbm_on) else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(lib.ddp_segment_reduce(_ptr(x), ldx, n_nodes, d_out, arr, len(sources), 1 if accumulate else 0, _stream()),
            "ddp_segment_reduce")
    if prof is not None:
        e1.record()
        # algorithmic bytes: every message row read once, every node row read and written once
        ne = sum(s_[1].n_edges for s_ in sources)
        prof.hbm.setdefault("ddp_segment_reduce_kernel", []).append((e0, e1, 4.0 * d_out * (ne + 2 * n_nodes)))


class _EdgeMLPPack:
    """Host-side split of an edge-embedding MLP `Linear(in, ns) -> ReLU -> Linear(ns, ns)` for ddp_edge_featurize:
    the RBF columns of the first Linear go to the kernel (zero padded to 64 outputs); the other input columns
    (sigma embedding, bond type) plus the bias become the per-node / per-edge `pre` table."""

    def __init__(self, seq: nn.Sequential, rbf_slice: slice, device):
        W1, b1, W2, b2 = seq[0].weight.detach(), seq[0].bias.detach(), seq[3].weight.detach(), seq[3].bias.detach()
        ns = W1.shape[0]
        k = rbf_slice.stop - rbf_slice.start
        w1d = torch.zeros(k, 64, device=device)
        w1d[:, :ns] = W1[:, rbf_slice].t()
        w2 = torch.zeros(64, 64, device=device)
        w2[:ns, :ns] = W2.t()
        b2p = torch.zeros(64, device=device)
        b2p[:ns] = b2
        self.w1d, self.w2, self.b2, self.ns, self.k = w1d.contiguous(), w2.contiguous(), b2p, ns, k
        self.W1, self.b1 = W1, b1


def _edge_featurize(pack: _EdgeMLPPack, dist: GaussianSmearing, pos_a, ia, pos_b, ib, pre, pre_idx, pre2=None):
    """pre: [*, >= ns] rows with unit column stride (a column slice of a wider table is fine); pre2 (optional, [n2, ns]) is
    added to the first n2 edges' rows (the bond-type columns of lig_edge_embedding's first Linear)."""
    lib = L.load()
    E = int(ia.shape[0])
    dev = pos_a.device
    out = torch.empty((E, pack.ns), device=dev, dtype=torch.float32)
    sh = torch.empty((E, 4), device=dev, dtype=torch.float32)
    if E == 0:
        return out, sh
    if pre.stride(1) != 1:
        pre = pre.contiguous()
    n2 = 0 if pre2 is None else int(pre2.shape[0])
    L.check(lib.ddp_edge_featurize(_ptr(pos_a), _ptr(ia), _ptr(pos_b), _ptr(ib), E, None, _ptr(dist.offset), pack.k,
                                   C.c_float(dist.coeff), _ptr(pre), _ptr(pre_idx), pre.stride(0),
                                   _ptr(pre2) if n2 else None, n2, pre2.stride(0) if n2 else 0, _ptr(pack.w1d),
                                   _ptr(pack.w2), _ptr(pack.b2), pack.ns, _ptr(out), _ptr(sh), _stream()),
            "ddp_edge_featurize")
    return out, sh


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


def _node_job(n_rows, out, ncols, w, bias, zero_to=0, cat=None, pack=None, emb_mode=0, dense=(), sigma=None, sig_out=None):
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
                       "smooth_edges": smooth_edges, "odd_parity": odd_parity,
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
        # Both plans above (and the dead-output walk) cost host time - ~100 small launches and a few synchronisations, 2.7 ms
        # for the walk - that is hidden behind the conv layers of a large batch but sits on the critical path of a small one
        # (5 samples of 3dpf: 9.9 ms per step, host bound).  They pay when the layers they run behind take longer than they
        # do: measured cross-over at ~12 samples of 3dpf, expressed in atom-atom edges so that it scales with the complex
        self.plan_min_edges = 100_000
        self._static_cache = {}        # see _cached()
        self.prune_async = True        # dead-output walk on a side stream behind the first layers (forward)
        self._side = None
        self.before_layers = None      # optional callable, run once per forward between the front (graphs, edge embeddings,
                                       # CSR views) and the conv layers: sampler.PipelinedSampler orders the layers of its
                                       # resident groups with it (an event wait on the current stream)
        self.cache_slot = 0            # callers that alternate between several resident batches (sampler.PipelinedSampler)
                                       # give each its own slot so that they do not evict each other's entries
        self._stage_a_stacks = {}      # (layer, conv ids) -> stacked stage-A right-hand sides, see _stage_a()
        self.section_timer = None      # optional SectionTimer (tools/time_sections.py): per-section GPU + host time
        self.check_weight_values = True  # see _refresh_weight_caches
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

    # ---- checkpoint compatibility -------------------------------------------------------------
    _IGNORED_PREFIXES = ("final_tp_tor.", "final_tp_sc_tor.", "tor_bond_conv.tp.", "sc_tor_bond_conv.tp.")

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Reference checkpoints carry e3nn-internal buffers under final_tp_tor.* / *.tp.* (SURVEY §8(c)); they hold
        no learnable state and are dropped."""
        sd = {k: v for k, v in state_dict.items() if not k.startswith(self._IGNORED_PREFIXES)}
        out = super().load_state_dict(sd, strict=strict, **kw)
        self.invalidate_packed()
        return out

    def _stage_a(self, l, convs, x_src):
        """Stage A of the factorised convs of layer `l` that read the same source-node array: ONE ddp_stage_a launch
        rows[(conv, slot)] = x_src[:, scalars(slot)] @ Wg[(conv, slot)] = [G | Gb | pad] for all of them (weight-stationary
        fp32-MFMA kernel, csrc/ddp_gemm.hip; bound by the HBM write of G).  convs: [(k, TensorProductConvLayer)].
        Returns {(k, slot): G rows}."""
        lib = L.load()
        key = (l, tuple(k for k, _ in convs))
        ent = self._stage_a_stacks.get(key)
        if ent is None or ent[0].device != x_src.device:
            Ws, meta = [], []
            for k, conv in convs:
                pk = conv.packed_g(x_src.device)
                for slot in (0, 1):
                    if pk.wg[slot] is not None:
                        Ws.append(pk.wg[slot])
                        meta.append((k, slot, pk.g_in_off[slot]))
            ent = (torch.stack(Ws).contiguous(), meta, (C.c_int32 * len(meta))(*[m[2] for m in meta]))
            self._stage_a_stacks[key] = ent
        Wst, meta, offs = ent
        nb, N, n_in = len(meta), x_src.shape[0], Wst.shape[1]
        if nb > L.DDP_MAX_GEMM_BATCH:
            raise L.DdpError("more (conv, slot) pairs per source array than DDP_MAX_GEMM_BATCH")
        Gall = torch.empty((nb, N, Wst.shape[2]), device=x_src.device, dtype=torch.float32)   # 128-byte aligned rows
        prof = _PROFILER if (_PROFILER is not None and _PROFILER.hbm_on) else None
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(lib.ddp_stage_a(x_src.data_ptr(), x_src.shape[1], N, offs, nb, Wst.data_ptr(), n_in, Wst.shape[2],
                                Gall.data_ptr(), Wst.shape[2], _stream()), "ddp_stage_a")
        if prof is not None:
            e1.record()
            # algorithmic bytes: the G rows written once + the scalar columns of x read once per product + the weights
            prof.hbm.setdefault("ddp_stage_a_mfma_kernel", []).append(
                (e0, e1, 4.0 * (Gall.numel() + nb * N * n_in + Wst.numel())))
        return {(k, slot): Gall[i] for i, (k, slot, _) in enumerate(meta)}

    def _shared_receptor_side(self, B, rec, atom, rpos, apos, lay_r, lay_a, rr, ar, aa, atoms=True):
        """Which receptor-side convs see the SAME problem in every graph of the batch (the usual sampling batch: N poses of
        one complex): per conv k in (3 atom<-atom, 5 atom<-receptor, 6 receptor<-receptor, 8 receptor<-atom) either None
        or (receiver nodes per graph, edges per graph, source nodes per graph).  Exact comparison of node features,
        positions and per-graph edge lists; depends only on step-independent tensors, so it is evaluated once (`_cached`).
        atoms=False: the atom side is not examined (flexible side chains move per sample and per step: its comparison
        would fail anyway, after a handful of host synchronisations on every call)."""
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
        atom_same = atoms and same_rows(atom.x, na) and same_rows(apos, na)
        if rec_same:
            e = same_edges(rr, nr, nr)
            out[6] = (nr, e, nr) if e else None
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

    def invalidate_packed(self):
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

    def _side_stream(self, dev):
        if self._side is None or self._side.device != torch.device(dev):
            self._side = torch.cuda.Stream(device=dev, priority=-1)
        return self._side

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
                jobs.append(_node_job(N, x, ns, pk.w, pk.b, zero_to=ldx, cat=cat, pack=pk, emb_mode=1,
                                      dense=[(xf, ncat, n_lm)] if n_lm else [], sigma=sigma, sig_out=sig_out))
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
    @torch.no_grad()
    def forward(self, data):
        lig, rec, atom = data["ligand"], data["receptor"], data["atom"]
        _require_hip(lig.pos)
        dev = lig.pos.device
        self._refresh_weight_caches()
        ns, L_ = self.ns, self.num_conv_layers
        B = int(data.num_graphs)
        mark = self.section_timer.mark if self.section_timer is not None else (lambda name: None)
        mark("start")
        if self.no_aminoacid_identities:
            rec.x = rec.x * 0
        if self.confidence_mode:   # (:245) the times are used as they are
            tr_sigma, rot_sigma, tor_sigma, sc_sigma = [data.complex_t[k] for k in ("tr", "rot", "tor", "sc_tor")]
        else:
            tr_sigma, rot_sigma, tor_sigma, sc_sigma = self.t_to_sigma(*[data.complex_t[k] for k in ("tr", "rot", "tor", "sc_tor")])

        lpos, rpos, apos = lig.pos.float().contiguous(), rec.pos.float().contiguous(), atom.pos.float().contiguous()
        lbatch, rbatch, abatch = lig.batch.long(), rec.batch.long(), atom.batch.long()
        Nl, Nr, Na = lpos.shape[0], rpos.shape[0], apos.shape[0]
        lay_l = self._cached("lay_l", (lbatch,), lambda: G.DenseLayout.build(lbatch, B))
        lay_r = self._cached("lay_r", (rbatch,), lambda: G.DenseLayout.build(rbatch, B))
        lay_a = self._cached("lay_a", (abatch,), lambda: G.DenseLayout.build(abatch, B))

        # node encoders, sigma embeddings and the per-node part of the edge-embedding MLPs' first Linear: one HIP launch,
        # queued ahead of the searches' host synchronisation (it depends on the diffusion time and the node features only)
        ldx = self._ldx
        xl, xr, xa, pre = self._node_tables(lig, rec, atom, dev)
        mark("node_embed")
        # ---- graphs (:444-583)
        i32 = lambda t: t.to(torch.int32).contiguous()
        bond_ei = data["ligand", "ligand"].edge_index.long()
        sd_, dd, cd = self.sigma_embed_dim, self.distance_embed_dim, self.cross_distance_embed_dim
        nf = self.in_lig_edge_features
        epk = {}
        for key, name, rbf0, rbf_n in (("ll", "lig_edge_embedding", nf + sd_, dd), ("rr", "rec_edge_embedding", sd_, dd),
                                       ("aa", "atom_edge_embedding", sd_, dd), ("lr", "lr_edge_embedding", sd_, cd),
                                       ("la", "la_edge_embedding", sd_, cd), ("ar", "ar_edge_embedding", sd_, dd)):
            epk[key] = self._edge_pack(name, slice(rbf0, rbf0 + rbf_n), dev)
        # bond-type columns of lig_edge_embedding's first Linear, [E_bond, ns]: fixed for a batch
        bond_attr = data["ligand", "ligand"].edge_attr
        bond_pre = self._cached("bond_pre", (bond_attr,), lambda: bond_attr.float() @ epk["ll"].W1[:, :nf].t())

        # The neighbour searches that depend on the pose - ligand radius graph, ligand<-receptor, ligand<-atom and the heads'
        # bond-centre graphs - are COUNTED first and share one host synchronisation (G.resolve); the heads' graphs are
        # filled only after the conv layers are queued (nothing before the layers needs them unless side chains are flexible)
        s_ll = G.RadiusSearch.graph(lpos, self.lig_max_radius, lay_l)
        rr = data["receptor", "receptor"].edge_index.long()
        aa = self._cached("aa", (apos, abatch), lambda: G.knn_graph(apos, self.atom_max_neighbors if self.atom_max_neighbors else 32, lay_a))
        data["atom", "atom"].edge_index = aa
        if self.dynamic_max_cross:
            cut = (tr_sigma * 3 + 20).unsqueeze(1)
            s_lr = G.RadiusSearch(rpos / cut[rbatch], lpos / cut[lbatch], 1.0, lay_r, lay_l, max_num_neighbors=10000)
        else:
            s_lr = G.RadiusSearch(rpos, lpos, self.cross_max_distance, lay_r, lay_l, max_num_neighbors=10000)
        s_la = G.RadiusSearch(apos, lpos, self.lig_max_radius, lay_a, lay_l, max_num_neighbors=10000)
        # the atoms that occur as sources of ligand<-atom edges = the atoms with a ligand atom of their graph within the
        # radius: marked by the same search with the roles swapped (count pass only: 37 candidates per atom), so that their
        # NUMBER rides in the searches' one host synchronisation and the compact source numbering of that conv (below) needs
        # no synchronisation of its own (it was a torch.unique_consecutive: a second wait in the host-paced front)
        near_atom = None
        if self.factorize_min_degree > 0:
            near_atom = G.RadiusSearch(lpos, apos, self.lig_max_radius, lay_l, lay_a, max_num_neighbors=10000).counts > 0
        num_flex = 0
        # (:327, literally: a PyG HeteroData answers `in` by attribute names, not node types, and creates the store on access)
        if self.flexible_sidechains and len(data["flexResidues"]) > 0:
            num_flex = int(data["flexResidues"].edge_idx.shape[0])
        pend_tor = pend_sc = rot_bond_idx = None
        if not self.confidence_mode:
            if not self.no_torsion:
                def tor_static():   # rotatable bonds, their graph index and dense layout: fixed for a batch
                    idx = lig.edge_mask.bool().nonzero(as_tuple=True)[0]
                    bnd = bond_ei[:, idx]
                    bb = lbatch[bnd[0]]
                    return idx, bnd, bb, (G.DenseLayout.build(bb, B) if idx.shape[0] > 0 else None)

                rot_bond_idx, bonds_t, bond_batch, lay_b = self._cached("tor_static", (lig.edge_mask, bond_ei, lbatch), tor_static)
                if rot_bond_idx.shape[0] > 0:
                    pend_tor = self._torsion_search("final_edge_embedding", lpos, lay_l, bonds_t, bond_batch, B, lay_b)
            if num_flex > 0:
                fr = data["flexResidues"]
                bonds_s = lay_a.starts[fr.batch.long()] + fr.edge_idx.t().long()     # get_sc_tor_bonds (:638-652)
                sc_batch = fr.batch.long()
                lay_sc = self._cached("lay_sc", (fr.batch,), lambda: G.DenseLayout.build(sc_batch, B))   # (its build syncs)
                pend_sc = self._torsion_search("sidechain_final_edge_embedding", apos, lay_a, bonds_s, sc_batch, B, lay_sc)
        pending = [p for p in (pend_tor, pend_sc) if p is not None]
        # (the "all receptor-side nodes at one diffusion time" flag of the layer-0 sharing below rides in the same copy)
        flags = []
        if self.share_layer0 and B > 1:
            t_nodes = torch.cat([rec.node_t["tr"], atom.node_t["tr"]])
            flags.append((t_nodes == t_nodes[0]).all())
            if num_flex > 0 and lay_a.uniform:   # side chains usually differ between the samples: asked in the same copy
                av = apos.reshape(B, lay_a.nmax, 3)
                flags.append((av == av[:1]).all())
        mark("before_sync")
        extra = flags + ([near_atom.sum()] if near_atom is not None else [])
        counts = G.resolve([s_ll, s_lr, s_la] + [p["search"] for p in pending], extra=extra)
        mark("sync")
        n_near = counts.pop() if near_atom is not None else 0
        for p, e in zip(pending, counts[3:]):
            p["E"] = e
        one_time = bool(flags) and bool(counts[-len(flags)])
        atoms_alike = len(flags) < 2 or bool(counts[-1])
        ll = torch.cat([bond_ei, s_ll.fill(counts[0])], 1)
        lr = s_lr.fill(counts[1])
        la = s_la.fill(counts[2])
        ll32, lr32, la32 = (i32(ll[0]), i32(ll[1])), s_lr.row32, s_la.row32    # int32 rows for the kernels
        ar = data["atom", "receptor"].edge_index.long()
        self.last_stats = {"E_ll": ll.shape[1], "E_rr": rr.shape[1], "E_aa": aa.shape[1], "E_lr": lr.shape[1],
                           "E_la": la.shape[1], "E_ar": ar.shape[1], "N_l": Nl, "N_r": Nr, "N_a": Na, "B": B}

        # Batches of N poses of ONE complex at one diffusion time (the sampling loop): receptor-side quantities that do not
        # involve the ligand are the same in every graph.  Checked exactly (`_shared_receptor_side`, cached); used twice:
        #  * the edge embeddings / harmonics of receptor<-receptor, atom<-atom, atom<->receptor edges are computed for graph 0
        #    only and addressed through edge ids modulo the per-graph edge count;
        #  * layer 0 of those convs (before any message has been passed, receptor and atom features are just the node
        #    encoders' outputs) is computed for graph 0 only - its edges are a prefix of the receiver- and of the
        #    source-ordered edge lists - and the resulting node update is added to every graph.
        # Any difference between the graphs -> the general path.
        shared0 = {}
        dbg = self.debug_conv_outputs
        if dbg is not None:
            one_time = False
        if one_time and not atoms_alike:
            sh_ = self._cached("shared0_rec", (rec.x, rpos, rr),
                               lambda: self._shared_receptor_side(B, rec, atom, rpos, apos, lay_r, lay_a, rr, ar, aa, atoms=False))
            shared0 = {k: v for k, v in sh_.items() if v is not None}
        elif one_time:
            sh_ = self._cached("shared0", (rec.x, rpos, atom.x, apos, rr, ar, aa),
                               lambda: self._shared_receptor_side(B, rec, atom, rpos, apos, lay_r, lay_a, rr, ar, aa))
            shared0 = {k: v for k, v in sh_.items() if v is not None}

        def graph0(k, ei):   # canonical edges of graph 0 (a prefix: edge lists are graph-major) when conv k is shared
            return ei[:, :shared0[k][1]] if k in shared0 else ei

        rr_f, aa_f, ar_f = graph0(6, rr), graph0(3, aa), graph0(5, ar)

        def rows32(name, ei, ei_f):   # int32 rows of a step-independent edge set (kept across calls), cut like ei_f
            r0, r1 = self._cached(name, (ei,), lambda: (i32(ei[0]), i32(ei[1])))
            n = ei_f.shape[1]
            return r0[:n], r1[:n]

        rr32, aa32, ar32 = rows32("rr32", rr, rr_f), rows32("aa32", aa, aa_f), rows32("ar32", ar, ar_f)

        mark("graphs")
        # ---- edge featurisation (the per-node `pre` tables were prepared ahead of the searches, see above)
        e_ll, sh_ll = _edge_featurize(epk["ll"], self.lig_distance_expansion, lpos, ll32[0], lpos, ll32[1], pre["ll"], ll32[0],
                                      pre2=bond_pre)
        e_rr, sh_rr = _edge_featurize(epk["rr"], self.rec_distance_expansion, rpos, rr32[0], rpos, rr32[1], pre["rr"], rr32[0])
        e_aa, sh_aa = _edge_featurize(epk["aa"], self.lig_distance_expansion, apos, aa32[0], apos, aa32[1], pre["aa"], aa32[0])
        e_lr, sh_lr = _edge_featurize(epk["lr"], self.cross_distance_expansion, lpos, lr32[0], rpos, lr32[1], pre["lr"], lr32[0])
        e_la, sh_la = _edge_featurize(epk["la"], self.cross_distance_expansion, lpos, la32[0], apos, la32[1], pre["la"], la32[0])
        e_ar, sh_ar = _edge_featurize(epk["ar"], self.rec_distance_expansion, apos, ar32[0], rpos, ar32[1], pre["ar"], ar32[0])

        mark("edge_featurize")
        # ---- CSR per conv direction (receiver = edge_index[0] of the conv call)
        c_ll = G.build_csr(ll32[0], ll32[1], Nl)
        c_lr = G.build_csr(lr32[0], lr32[1], Nl, presorted=True)
        c_la = G.build_csr(la32[0], la32[1], Nl, presorted=True)
        def static_csr(name, k, ei, recv, src, n):
            """CSR view of a step-independent edge set, kept across calls; edge ids modulo the per-graph edge count when
            the edge embeddings exist for graph 0 only."""
            if k in shared0:
                e0 = shared0[k][1]

                def modded():
                    c = G.build_csr(recv, src, n)
                    return G.CSR(c.n_edges, c.recv, c.src, (c.eid % e0).contiguous(), c.rowptr)
                return self._cached(f"{name}_mod{e0}", (ei,), modded)
            return self._cached(name, (ei,), lambda: G.build_csr(recv, src, n))

        c_aa = static_csr("c_aa", 3, aa, aa[0], aa[1], Na)
        c_al = G.build_csr(la32[1], la32[0], Na)
        c_ar = static_csr("c_ar", 5, ar, ar[0], ar[1], Na)
        c_rr = static_csr("c_rr", 6, rr, rr[0], rr[1], Nr)
        c_rl = G.build_csr(lr32[1], lr32[0], Nr)
        c_ra = static_csr("c_ra", 8, ar, ar[1], ar[0], Nr)

        # conv k of a layer: (csr, receiver x, source x, edge_base, sh, receiver type)
        plan = [
            (0, c_ll, xl, xl, e_ll, sh_ll, "l"), (1, c_lr, xl, xr, e_lr, sh_lr, "l"), (2, c_la, xl, xa, e_la, sh_la, "l"),
            (3, c_aa, xa, xa, e_aa, sh_aa, "a"), (4, c_al, xa, xl, e_la, sh_la, "a"), (5, c_ar, xa, xr, e_ar, sh_ar, "a"),
            (6, c_rr, xr, xr, e_rr, sh_rr, "r"), (7, c_rl, xr, xl, e_lr, sh_lr, "r"), (8, c_ra, xr, xa, e_ar, sh_ar, "r"),
        ]
        nodes = {"l": (xl, Nl), "a": (xa, Na), "r": (xr, Nr)}
        # summation order of the residual update (:316,:320,:324): lig u0+u2+u1, atom u3+u4+u5, rec u6+u8+u7
        order = {"l": [0, 2, 1], "a": [3, 4, 5], "r": [6, 8, 7]}
        # source-ordered views for the factorised convs (built once per forward, reused by every layer)
        n_src_nodes = {"l": Nl, "a": Na, "r": Nr}
        src_type = {0: "l", 1: "r", 2: "a", 3: "a", 4: "l", 5: "r", 6: "r", 7: "l", 8: "a"}
        so_views, compact_src = {}, {}
        if self.factorize_min_degree > 0:
            for k, csr, *_ in plan:
                if csr.n_edges > 0 and csr.n_edges >= self.factorize_min_degree * n_src_nodes[src_type[k]]:
                    so_views[k] = self._cached(f"so_{k}", (csr.src, csr.eid), lambda: G.source_order(csr, n_src_nodes[src_type[k]])) if k in (3, 5, 6, 8) \
                        else G.source_order(csr, n_src_nodes[src_type[k]])
                elif k == 2 and csr.n_edges > 0:
                    # ligand<-atom: few edges per atom over ALL atoms, but the edges leave only the atoms around the ligand.
                    # Degree over the atoms that occur: factorise with stage A on those rows only (compact copy of x per layer)
                    if near_atom is not None and csr.n_edges >= self.factorize_min_degree * max(n_near, 1):
                        so_c = G.source_order(csr, Na)
                        uniq = torch.nonzero_static(near_atom, size=n_near).squeeze(1)          # ascending = source order
                        rank = torch.cumsum(near_atom, 0) - 1
                        so_views[k] = G.SourceOrder(so_c.n_edges, so_c.recv, rank[so_c.src.long()].to(torch.int32), so_c.eid, so_c.pos)
                        compact_src[k] = uniq
        # Layer 1, atom<-atom, sampling batches of one rigid complex (shared0 has conv 3): after the shared layer 0 an atom's
        # features differ between the samples only if an atom<-ligand message reached it (the atoms within 5 A of that
        # sample's ligand, ~15 %).  A layer-1 atom<-atom message between two atoms without such a message ("clean") is
        # therefore the same in every sample: those messages are computed ONCE on the complex's own edge list (e0 edges,
        # rows [E, E + e0) of the message array) and the segmented mean reads them through a row map; only the edges with a
        # touched end are computed per sample (source-ordered sub-list, stage A on their source rows only).  Messages of a
        # clean pair are bitwise those the general path computes (same inputs, per-edge arithmetic), the mean sums the same
        # values in the same order: the result is bitwise the general path's (GPU test).  Layer 1 must not be one of the pruned
        # last layers (L >= 4).
        clean1 = None
        clean1_on = bool(self.share_clean_layer1 and 3 in shared0 and 3 in so_views and L_ >= 4 and c_aa.n_edges > 0
                         and c_aa.n_edges >= self.plan_min_edges)

        def clean1_plan():
            n0_, e0_, _ = shared0[3]
            E_aa = c_aa.n_edges
            so3 = so_views[3]
            clean = (c_al.rowptr[1:] == c_al.rowptr[:-1]) if c_al.n_edges > 0 else torch.ones(Na, dtype=torch.bool, device=dev)
            dirty_so = ~(clean[so3.recv.long()] & clean[so3.src.long()])
            idx = dirty_so.nonzero(as_tuple=True)[0]                       # (host synchronisation: how many)
            n_d = int(idx.shape[0])
            self.last_stats["clean1_dirty_edges"] = n_d
            if n_d > 0.8 * E_aa:
                return None
            uniq_d = so_d = None
            if n_d > 0:
                src_d = so3.src[idx]
                uniq_d, inv = torch.unique_consecutive(src_d.long(), return_inverse=True)      # (second one)
                so_d = G.SourceOrder(n_d, so3.recv[idx].contiguous(), inv.to(torch.int32).contiguous(), so3.eid[idx].contiguous(),
                                     so3.pos[idx].contiguous())
            dirty_csr = ~(clean[c_aa.recv.long()] & clean[c_aa.src.long()])
            p_ = G.iota32(E_aa, dev)
            rowmap = torch.where(dirty_csr, p_, E_aa + p_ % e0_).to(torch.int32).contiguous()
            so_v = G.SourceOrder(e0_, so3.recv[:e0_], so3.src[:e0_], so3.eid[:e0_], (so3.pos[:e0_] + E_aa).contiguous())
            first = clean.view(B, n0_).to(torch.uint8).argmax(0)      # a sample in which the atom is clean (0 if none: unused)
            rows_v = first * n0_ + torch.arange(n0_, device=dev)
            return {"so_d": so_d, "uniq_d": uniq_d, "rowmap": rowmap, "so_v": so_v, "rows_v": rows_v, "E": E_aa, "e0": e0_}

        # The plan reads graph structure only and synchronises with the host twice: it runs on the side stream AFTER layer 0
        # is queued (like the dead-output walk below), so that the device works on layer 0 while the host waits for the sizes.
        # ---- graph parts of the heads.  Their only host synchronisation (the edge counts) happened above; with flexible
        # side chains the side-chain graph is completed here because the dead-output walk below reads it, everything else
        # (centre graph, ligand torsion graph) is queued after the conv layers, behind which it costs no wall time
        head_tor = None
        head_sc = self._torsion_finish(pend_sc, dev) if pend_sc is not None else None
        mark("head_sc_graph")
        # Dead-output elimination over the last layers.  What is read after the last layer: all ligand features (heads),
        # with flexible side chains the atom features around the flexible bonds (side-chain torsion head), nothing of the
        # receptor.  Walking backwards, a layer's receptor-side convs only have to produce the rows that are still read
        # (by the residual of a needed node or as source / receiver of a kept edge of the next layer), so their edge lists
        # are restricted to the edges that END in a needed node - exact, the other rows of x are simply left stale.
        # Without flexible side chains this prunes layer L-2 (its atom outputs feed only the final ligand<-atom conv), with
        # them layers L-1 and L-2.  The walk stops as soon as (almost) everything is needed; layer 0 is never touched.
        def prune_plan():
            pruned, pruned_so = {}, {}
            ALL = None
            need = {"l": ALL, "a": torch.zeros(Na, dtype=torch.bool, device=dev), "r": torch.zeros(Nr, dtype=torch.bool, device=dev)}
            if head_sc is not None:
                need["a"][head_sc["bonds"].reshape(-1)] = True
                need["a"][head_sc["csr"].src.long()] = True
            elif self.flexible_sidechains:
                need["a"] = ALL      # (flexible model without flexible residues in the batch: keep the general path)
            recv_of = {"a": (3, 4, 5), "r": (6, 7, 8)}
            for l in range(L_ - 1, 0, -1):
                act = {"l": True, "a": self.flexible_sidechains or l != L_ - 1}
                act["r"] = act["a"] and l != L_ - 1
                todo = [rt for rt in ("a", "r") if act[rt] and need[rt] is not ALL]
                if todo:
                    fr = torch.stack([need[rt].float().mean() for rt in todo]).tolist()
                    for rt, f in zip(todo, fr):
                        if f >= 0.85:        # e.g. the cross cutoff usually reaches every pocket residue
                            need[rt] = ALL
                cand = [(k, plan[k][1], rt) for rt in ("a", "r") if act[rt] and need[rt] is not ALL
                        for k in recv_of[rt] if plan[k][1].n_edges > 0]
                keeps = [need[rt][csr.recv.long()] for _, csr, rt in cand]
                counts = torch.stack([kp.sum() for kp in keeps]).tolist() if cand else []
                for (k, csr, rt), kp, e_keep in zip(cand, keeps, counts):
                    if e_keep >= 0.9 * csr.n_edges:      # not worth the re-indexing
                        continue
                    recv = csr.recv[kp]
                    n_rows = csr.rowptr.shape[0] - 1
                    cnts = torch.zeros(n_rows, dtype=torch.int64, device=dev).index_add_(0, recv.long(), torch.ones_like(recv, dtype=torch.int64))
                    rowptr = torch.zeros(n_rows + 1, dtype=torch.int32, device=dev)
                    rowptr[1:] = torch.cumsum(cnts, 0).to(torch.int32)
                    pruned.setdefault(l, {})[k] = G.CSR(int(e_keep), recv.contiguous(), csr.src[kp].contiguous(),
                                                        csr.eid[kp].contiguous(), rowptr)
                # rows of x(l) that layer l reads: the needed rows themselves (residual) and the sources of its kept edges
                nxt = {}
                for t in ("l", "a", "r"):
                    if need[t] is ALL:
                        nxt[t] = ALL
                        continue
                    m = need[t].clone()
                    for k, csr, _, _, _, _, rt in plan:
                        if act[rt] and src_type[k] == t and csr.n_edges > 0:
                            m[(pruned.get(l, {}).get(k, csr)).src.long()] = True
                    nxt[t] = m
                need = nxt
                if all(v is ALL for v in need.values()):
                    break
            # source-ordered views of the pruned factorised convs and, where few source nodes are left, their compact
            # numbering for stage A - all of it independent of the features, so done here (it synchronises with the host)
            for l, pl in pruned.items():
                for k, c in pl.items():
                    if k in so_views and c.n_edges > 0:
                        so_p, uniq = G.source_order(c, n_src_nodes[src_type[k]]), None
                        if c.n_edges * 2 < plan[k][1].n_edges:
                            uniq, inv = torch.unique_consecutive(so_p.src.long(), return_inverse=True)
                            so_p = G.SourceOrder(so_p.n_edges, so_p.recv, inv.to(torch.int32), so_p.eid, so_p.pos)
                        pruned_so.setdefault(l, {})[k] = (so_p, uniq)
            return pruned, pruned_so

        # The walk costs ~1.7 ms of small launches with host synchronisations (data-dependent sizes) and only the last layers
        # need its result: with `prune_async` it runs on a side stream AFTER layers 0 .. L-4 are queued - its inputs are graph
        # structure only, not features - so that it hides behind those layers instead of delaying the first one.  A layer
        # that is already queued when the plan arrives simply runs unpruned (always exact).
        pruned, pruned_so = {}, {}
        prune_on = (self.prune_last_receptor_layer and L_ >= 2 and not self.confidence_mode
                    and c_aa.n_edges >= self.plan_min_edges and dbg is None)
        prune_after = (L_ - 4) if (prune_on and self.prune_async and L_ >= 4 and self.before_layers is None) else None
        if prune_on and prune_after is None:
            pruned, pruned_so = prune_plan()
        inputs_ready = None
        if prune_after is not None or clean1_on:
            inputs_ready = torch.cuda.Event()
            inputs_ready.record()
        mark("csr")
        if self.before_layers is not None:
            self.before_layers()
        for l in range(L_):
            spec, spec_g = self._layer_specs[l], self._layer_specs_g[l]
            do_atom = self.flexible_sidechains or l != L_ - 1
            do_rec = do_atom and l != L_ - 1
            active = {"l": True, "a": do_atom, "r": do_rec}
            shared = shared0 if l == 0 else {}
            if prune_after is not None and l == prune_after + 1:
                side = self._side_stream(dev)
                side.wait_event(inputs_ready)
                with torch.cuda.stream(side):
                    pruned, pruned_so = prune_plan()
                    plan_ready = torch.cuda.Event()
                    plan_ready.record()
                torch.cuda.current_stream().wait_event(plan_ready)
                pruned = {ll_: v for ll_, v in pruned.items() if ll_ > prune_after}
                pruned_so = {ll_: v for ll_, v in pruned_so.items() if ll_ > prune_after}
                mark("prune_plan")
            if l == 1 and clean1_on:
                if self.before_layers is None:
                    side = self._side_stream(dev)
                    side.wait_event(inputs_ready)
                    with torch.cuda.stream(side):
                        clean1 = clean1_plan()
                        c1_ready = torch.cuda.Event()
                        c1_ready.record()
                    torch.cuda.current_stream().wait_event(c1_ready)
                else:   # (resident sample groups order their streams themselves: plan on the current stream)
                    clean1 = clean1_plan()
            layer_csr, layer_so = pruned.get(l, {}), pruned_so.get(l, {})
            tasks, tasks_g, msgs, keep = [], [], {}, []
            # per conv of this layer: (csr, source-ordered view or None, source-node array) after the layer-specific
            # restrictions (pruned last receptor layer, graph-0 prefix of shared layer-0 convs)
            per = {}
            for k, csr, x_recv, x_src, e_base, sh, rt in plan:
                if not active[rt]:
                    continue
                so_k, xs_k = so_views.get(k), x_src
                if k in compact_src:
                    xs_k = x_src.index_select(0, compact_src[k])
                if k in layer_csr:   # edges that end in a node the final layer reads
                    csr = layer_csr[k]
                    if so_k is not None:
                        so_k, uniq = layer_so.get(k, (None, None))
                        if uniq is not None:   # stage A on the remaining source rows only (compact copy of x, renumbered src)
                            xs_k = x_src.index_select(0, uniq)
                if k in shared:   # graph 0's edges = a prefix of both orderings
                    n0, e0, _ = shared[k]
                    csr = G.CSR(e0, csr.recv[:e0], csr.src[:e0], csr.eid[:e0], csr.rowptr[:n0 + 1])
                    if so_k is not None:
                        so_k = G.SourceOrder(e0, so_k.recv[:e0], so_k.src[:e0], so_k.eid[:e0], so_k.pos[:e0])
                    xs_k = x_src[:shared[k][2]]
                per[k] = (csr, so_k, xs_k)
            c1 = clean1 if (l == 1 and clean1 is not None and 3 in per and 3 not in layer_csr) else None
            if c1 is not None:   # atom<-atom at layer 1: per-sample part = the edges with a touched end (see above)
                xs_d = xa.index_select(0, c1["uniq_d"]) if c1["uniq_d"] is not None else xa[:0]
                per[3] = (per[3][0], c1["so_d"], xs_d)
                x_clean = xa.index_select(0, c1["rows_v"])
                keep.append((xs_d, x_clean))
            keep.append(per)   # the launches below take raw pointers: per-layer views must outlive them
            # stage A of the factorised convs: one batched product per source-node array
            gmap, groups = {}, {}
            for k, (csr_k, so_k, xs_k) in per.items():
                if so_k is not None and csr_k.n_edges > 0 and so_k.n_edges > 0:
                    groups.setdefault((xs_k.data_ptr(), xs_k.shape[0]), (xs_k, []))[1].append((k, self.conv_layers[9 * l + k]))
            for xs_k, grp in groups.values():
                gmap.update(self._stage_a(l, grp, xs_k))
            if c1 is not None:
                gmap.update({("v", sl): g_ for (_, sl), g_ in self._stage_a(l, [(3, self.conv_layers[9 * l + 3])], x_clean).items()})
            keep.append(gmap)
            for k, csr, x_recv, x_src, e_base, sh, rt in plan:
                if not active[rt]:
                    continue
                conv = self.conv_layers[9 * l + k]
                pkc = conv.packed(dev)
                csr, so_k, xs_k = per[k]
                if k == 3 and c1 is not None:
                    # rows [0, E): per-sample messages at their CSR positions (only the touched edges are written and read),
                    # rows [E, E + e0): the messages of the complex's own edge list between clean atoms
                    msg = torch.empty((c1["E"] + c1["e0"], spec.d_out), device=dev, dtype=torch.float32)
                    msgs[k] = (msg, csr, pkc, c1["rowmap"])
                    pkg = conv.packed_g(dev)
                    if so_k is not None and so_k.n_edges > 0:
                        g = [gmap.get((k, sl)) for sl in (0, 1)]
                        segs = [(e_base, so_k.eid, ns, ns), (x_recv, so_k.recv, ldx, ns), (xs_k, so_k.src, ldx, ns)]
                        tasks_g.append(_make_task(pkg, xs_k, ldx, so_k, sh, segs, msg, g=g, pos=so_k.pos))
                    sv = c1["so_v"]
                    g = [gmap.get(("v", sl)) for sl in (0, 1)]
                    segs = [(e_base, sv.eid, ns, ns), (x_clean, sv.recv, ldx, ns), (x_clean, sv.src, ldx, ns)]
                    tasks_g.append(_make_task(pkg, x_clean, ldx, sv, sh, segs, msg, g=g, pos=sv.pos))
                    continue
                msg = torch.empty((csr.n_edges, spec.d_out), device=dev, dtype=torch.float32)
                msgs[k] = (msg, csr, pkc)
                if csr.n_edges == 0:
                    continue
                if so_k is not None:
                    so, pkg = so_k, conv.packed_g(dev)
                    g = [gmap.get((k, sl)) for sl in (0, 1)]
                    segs = [(e_base, so.eid, ns, ns), (x_recv, so.recv, ldx, ns), (xs_k, so.src, ldx, ns)]
                    tasks_g.append(_make_task(pkg, xs_k, ldx, so, sh, segs, msg, g=g, pos=so.pos))
                else:
                    segs = [(e_base, csr.eid, ns, ns), (x_recv, csr.recv, ldx, ns), (x_src, csr.src, ldx, ns)]
                    tasks.append(_make_task(pkc, x_src, ldx, csr, sh, segs, msg))
            nb_g = nb_d = 0.0
            if _PROFILER is not None:   # algorithmic node bytes of the conv calls of this layer (profiler only)
                d_in = P.irreps_dim(P.irreps_muls(ns, self.nv, l))
                for k, (csr_k, so_k, xs_k) in per.items():
                    if csr_k.n_edges > 0:
                        b_ = 4.0 * (nodes[src_type[k]][1] * d_in + nodes[plan[k][6]][1] * spec.d_out)
                        if so_k is not None:
                            nb_g += b_
                        else:
                            nb_d += b_
            mark("conv_prep")
            # (measured without gain on one MI355X: stage A on a second stream beside the direct convs, and the direct
            # convs on a second stream beside the factorised ones - neither pair fits on a CU together)
            _launch_convs(spec_g, tasks_g, flops_spec=spec, node_bytes=nb_g)
            _launch_convs(spec, tasks, node_bytes=nb_d)
            mark("conv_launch")
            if dbg is not None:   # every conv's own output = segmented mean + BatchNorm of its messages alone
                for k, ent in msgs.items():
                    n_k = nodes[plan[k][6]][1]
                    if ent[1].n_edges == 0:
                        dbg[f"conv_layers.{9 * l + k}"] = torch.zeros((), device=dev)
                        continue
                    o_k = torch.zeros((n_k, spec.d_out), device=dev)
                    _launch_reduce(o_k, spec.d_out, n_k, spec.d_out, [ent], accumulate=False)
                    dbg[f"conv_layers.{9 * l + k}"] = o_k
            for rt in ("l", "a", "r"):
                if active[rt]:
                    x, n = nodes[rt]
                    own = [msgs[k] for k in order[rt] if k not in shared]
                    if own:
                        _launch_reduce(x, ldx, n, spec.d_out, own, accumulate=True)
                    com = [msgs[k] for k in order[rt] if k in shared]
                    if com:   # graph 0's update of the shared convs, added to every graph
                        n0 = shared[[k for k in order[rt] if k in shared][0]][0]
                        u0 = torch.empty((n0, spec.d_out), device=dev, dtype=torch.float32)
                        _launch_reduce(u0, spec.d_out, n0, spec.d_out, com, accumulate=False)
                        x.view(B, n0, ldx)[:, :, :spec.d_out].add_(u0)   # (`+=` on the slice would copy the sum onto itself)
            mark("reduce")

        if self.confidence_mode:   # (:329-353) mean of the scalar channels per graph -> MLP
            def scalars(x):
                return torch.cat([x[:, :ns], x[:, self._d_final - ns:self._d_final]], dim=1) if L_ >= 3 else x[:, :ns]

            def graph_mean(v, b):
                out = torch.zeros((B, v.shape[1]), device=dev).index_add_(0, b, v)
                return out / torch.bincount(b, minlength=B).clamp(min=1).unsqueeze(1)

            conf_in = lay_l.dense(scalars(xl), 0.0).sum(1) / lay_l.counts.clamp(min=1).unsqueeze(1)   # (deterministic order)
            if self.flexible_sidechains:
                if num_flex > 0:
                    fr = data["flexResidues"]
                    bonds = lay_a.starts[fr.batch.long()] + fr.edge_idx.t().long()
                    flex_atoms = torch.unique(bonds)
                    conf_in = torch.cat([conf_in, graph_mean(scalars(xa)[flex_atoms], abatch[flex_atoms])], dim=1)
                else:
                    conf_in = torch.cat([conf_in, torch.zeros_like(conf_in)], dim=1)
            return self.confidence_predictor(conf_in).squeeze(dim=-1)

        # ---- graph parts of the remaining heads (sync-free, see above)
        ar_l = torch.arange(Nl, device=dev)
        cnt = lay_l.counts.unsqueeze(1)
        # (a dense sum, not index_add_: float atomics land in an order that depends on what else the device is doing, and
        # one ulp in the centre is one ulp in tr / rot - seen as run-to-run differences at the full size)
        center = lay_l.dense(lpos, 0.0).sum(1) / cnt
        pk = self._edge_pack("center_edge_embedding", slice(0, dd), dev)
        e_c, sh_c = _edge_featurize(pk, self.center_distance_expansion, center, i32(lbatch), lpos, i32(ar_l), pre["center"], i32(ar_l))
        c_c = self._cached("c_c", (lbatch,), lambda: G.build_csr(lbatch, ar_l, B, presorted=True))
        if pend_tor is not None:
            head_tor = self._torsion_finish(pend_tor, dev)
        mark("head_graphs")
        # ---- translation / rotation head (:357-384)
        fspec = self.final_conv.spec
        pkc = self.final_conv.packed(dev)
        msg = torch.empty((Nl, fspec.d_out), device=dev)
        seg_idx = c_c.src if self.fixed_center_conv else c_c.recv
        _launch_convs(fspec, [_make_task(pkc, xl, ldx, c_c, sh_c, [(e_c, c_c.eid, ns, ns), (xl, seg_idx, ldx, ns)], msg)])
        gp = torch.zeros((B, fspec.d_out), device=dev)
        _launch_reduce(gp, fspec.d_out, B, fspec.d_out, [(msg, c_c, pkc)], accumulate=False)
        if dbg is not None:
            dbg["final_conv"] = gp
        tr_pred = gp[:, :3] + gp[:, 6:9]
        rot_pred = gp[:, 3:6] + gp[:, 9:]
        data.graph_sigma_emb = self.timestep_emb_func(data.complex_t["tr"])
        tr_norm = torch.linalg.vector_norm(tr_pred, dim=1).unsqueeze(1)
        tr_pred = tr_pred / tr_norm * self.tr_final_layer(torch.cat([tr_norm, data.graph_sigma_emb], dim=1))
        rot_norm = torch.linalg.vector_norm(rot_pred, dim=1).unsqueeze(1)
        rot_pred = rot_pred / rot_norm * self.rot_final_layer(torch.cat([rot_norm, data.graph_sigma_emb], dim=1))
        if self.scale_by_sigma:
            tr_pred = tr_pred / tr_sigma.unsqueeze(1)
            rot_pred = rot_pred * self._so3_score_norm(rot_sigma).unsqueeze(1)

        mark("center_head")
        # ---- torsion heads (:386-434)
        if head_tor is None:
            tor_pred = torch.empty(0, device=dev)
        else:
            tor_pred = self._torsion_apply(self.tor_bond_conv, self.tor_final_layer, head_tor, xl, dev, "tor_bond_conv")
            if self.scale_by_sigma:
                edge_sigma = tor_sigma[lbatch][bond_ei[0]][rot_bond_idx]
                tor_pred = tor_pred * torch.sqrt(self._torus_score_norm(edge_sigma))
        if head_sc is None:
            sc_pred = torch.empty(0, device=dev)
        else:
            sc_pred = self._torsion_apply(self.sc_tor_bond_conv, self.sc_tor_final_layer, head_sc, xa, dev, "sc_tor_bond_conv")
            if self.scale_by_sigma:
                sc_pred = sc_pred * torch.sqrt(self._torus_score_norm(sc_sigma[data["flexResidues"].batch.long()]))
        mark("tor_heads")
        return tr_pred, rot_pred, tor_pred, sc_pred

    def _torsion_search(self, edge_mlp_name, pos, lay, bonds, bond_batch, B, lay_b=None):
        """First half of build_bond_conv_graph / build_sidechain_conv_graph (:586-636): the bond-centre radius search is
        counted (its edge count is read by forward() together with the other searches' - one host synchronisation)."""
        bond_pos = ((pos[bonds[0]] + pos[bonds[1]]) / 2).contiguous()
        if lay_b is None:
            lay_b = G.DenseLayout.build(bond_batch, B)
        return {"name": edge_mlp_name, "pos": pos, "bonds": bonds, "bond_pos": bond_pos,
                "search": G.RadiusSearch(pos, bond_pos, self.lig_max_radius, lay, lay_b)}   # [bond; atom], default cap 32

    def _torsion_finish(self, tg, dev):
        """Second half: everything of a torsion head that depends only on positions - the edges of the bond-centre graph,
        their embedding and harmonics, the CSR view.  No host synchronisation: forward() queues it after the conv layers."""
        lib = L.load()
        pos, bonds, bond_pos, E = tg["pos"], tg["bonds"], tg["bond_pos"], tg["E"]
        T = bonds.shape[1]
        if E == 0:
            raise RuntimeError("torsion head has no edges (the reference fails here as well)")
        tg["search"].fill(E)
        r32 = tg["search"].row32            # [bond; atom] int32
        pk = self._edge_pack(tg["name"], slice(0, self.distance_embed_dim), dev)
        pre = pk.b1.reshape(1, -1).contiguous()
        e_t, sh_e = _edge_featurize(pk, self.lig_distance_expansion, bond_pos, r32[0], pos, r32[1], pre,
                                    torch.zeros(E, device=dev, dtype=torch.int32))
        bond_vec = (pos[bonds[1]] - pos[bonds[0]]).contiguous()
        tor_sh = torch.empty((E, 4), device=dev)
        bond_of_edge = r32[0]
        L.check(lib.ddp_torsion_sh(_ptr(sh_e), _ptr(bond_vec), _ptr(bond_of_edge), E, _ptr(tor_sh), _stream()), "ddp_torsion_sh")
        csr = G.build_csr(r32[0], r32[1], T, presorted=True)
        return {"bonds": bonds, "T": T, "E": E, "e_t": e_t, "tor_sh": tor_sh, "csr": csr}

    def _torsion_apply(self, conv: TensorProductConvLayer, final_layer, tg, x, dev, name=None):
        """FullTensorProduct + tor_bond_conv + final layer (:386-434) on the graph of _torsion_graph; no host sync."""
        ns, ldx = self.ns, self._ldx
        bonds, csr, T, E = tg["bonds"], tg["csr"], tg["T"], tg["E"]
        bond_attr = (x[bonds[0], :ns] + x[bonds[1], :ns]).contiguous()
        spec, pkc = conv.spec, conv.packed(dev)
        msg = torch.empty((E, spec.d_out), device=dev)
        segs = [(tg["e_t"], csr.eid, ns, ns), (x, csr.src, ldx, ns), (bond_attr, csr.recv, ns, ns)]
        _launch_convs(spec, [_make_task(pkc, x, ldx, csr, tg["tor_sh"], segs, msg)])
        h = torch.zeros((T, spec.d_out), device=dev)
        _launch_reduce(h, spec.d_out, T, spec.d_out, [(msg, csr, pkc)], accumulate=False)
        if self.debug_conv_outputs is not None and name:
            self.debug_conv_outputs[name] = h
        return final_layer(h).squeeze(1)
