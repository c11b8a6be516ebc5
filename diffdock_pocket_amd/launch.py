"""Thin host wrappers around the C ABI of libddp_hip.so (include/ddp_hip.h) for the score-model forward: conv / reduce /
featurise / stage-A launches and the batched index-list primitives with device-side counts.

Every launch goes to the CURRENT stream of the CURRENT device (one C call to fetch the raw handle); the forward runs inside
`torch.cuda.device(batch device)`, so the searches, list kernels, convs and the pose update of one model share one stream
whatever the process-wide current device is.  Nothing here synchronises with the host.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from . import _lib as L
from . import packing as P
from .graph import EdgeView


def require_hip(t: torch.Tensor):
    if not t.is_cuda:
        raise L.DdpError("the MI355X score model runs on a HIP device only (no CPU/eager fallback); "
                         "move the batch to cuda:<n>")
    L.load()


def stream():
    """Raw handle of the current HIP stream of the current device (torch.cuda.current_stream() is ~9 us of Python)."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _p(t: Optional[torch.Tensor]) -> int:
    return t.data_ptr() if t is not None else 0


# ------------------------------------------------------------------------------------------------ device-side counts
class CountBlock:
    """The device-side sizes of one forward's pose-dependent lists: named int32 slots of one small tensor.  `cnt[name]` is a
    1-element view (allocated on first use) whose address goes to the kernels; nothing reads it on the host unless asked
    (`value`, used by last_stats / the profiler AFTER the step has been queued)."""

    SLOTS = 256

    def __init__(self, dev):
        self.block = torch.zeros(self.SLOTS, dtype=torch.int32, device=dev)
        self.index: Dict[str, int] = {}

    def __getitem__(self, name: str) -> torch.Tensor:
        i = self.index.get(name)
        if i is None:
            i = self.index[name] = len(self.index)
            if i >= self.SLOTS:
                raise L.DdpError("CountBlock: out of slots")
        return self.block[i:i + 1]

    def __contains__(self, name):
        return name in self.index

    def values(self) -> Dict[str, int]:
        """Host copy of every named slot (one device-to-host copy: a host synchronisation)."""
        host = self.block.tolist()
        return {k: int(host[i]) for k, i in self.index.items()}


# ------------------------------------------------------------------------------------------------ profiling hooks
class ConvProfiler:
    """Times every ddp_conv_messages launch with HIP events on the launch stream and tallies its algorithmic FLOPs
    (BASELINE.md section 3 formula x the launch's actual edge count).  Used by bench.py for the roofline entry.  Edge counts
    that live in device memory are resolved when a summary is asked for (after the timed region)."""

    def __init__(self):
        self.events, self.kernel, self.specs, self.counts, self.node_bytes, self.tags, self.h2 = [], [], [], [], [], [], []
        self.hbm = {}   # HBM-bound kernels: name -> [(event0, event1, bytes or callable)]
        self.hbm_on = False   # their ~45 extra event pairs per step cost wall time: bench.py times them in extra steps
        self._resolved = None

    # -- recording (called by the launch wrappers) -------------------------------------------------------------
    def record_conv(self, e0, e1, spec, flops_spec, tasks_counts, node_bytes, tag=None, h2=False, rows=False):
        """tasks_counts: [(capacity, cnt tensor or None)] of the launch's tasks; tag: where in the forward the launch sits
        ("layer3", "head")."""
        self.events.append((e0, e1))
        self.tags.append(tag)
        self.h2.append(bool(h2))     # the launch ran the fp16 hi/lo split form of the fc products
        # (rows: 1 = ddp_conv_rows_kernel, the v_mfma_f32_32x32x16_f16 form; 2 = ddp_conv_rows16_kernel, the 16x16x32 form)
        self.kernel.append((("ddp_conv_rows16_kernel" if spec.factorized else "ddp_conv_rows16_direct_kernel") if int(rows) == 2 else "ddp_conv_rows_kernel") if rows else
                           "ddp_conv32_kernel" if spec.factorized else "ddp_conv_messages_kernel")
        self.specs.append((spec, flops_spec or spec))
        self.counts.append(tasks_counts)
        self.node_bytes.append(node_bytes)
        self._resolved = None

    def _resolve(self):
        if self._resolved is None:
            torch.cuda.synchronize()
            cache = {}

            def val(cap, cnt):
                if cnt is None:
                    return cap
                key = (cnt.data_ptr(), id(cnt))
                if key not in cache:
                    cache[key] = min(cap, max(0, int(cnt.item())))
                return cache[key]

            ne = [sum(val(c, t) for c, t in tc) for tc in self.counts]
            self.edges = ne
            self.flops = [fs.flops_per_edge() * n for (s, fs), n in zip(self.specs, ne)]
            self.executed = [(s.mfma_flops_per_edge_executed() + 2 * s.hid * sum(s.g_cols)) * n for (s, fs), n in zip(self.specs, ne)]
            self.useful = [s.useful_flops_per_edge() * n for (s, fs), n in zip(self.specs, ne)]
            # product FLOPs that run as fp16 hi/lo split products: the two fc products; ddp_conv_rows also runs the per-edge G contraction
            # (h @ G[src]: 2 hid g_cols per edge) as tile products of the same form
            self.fc = [(s.fc_flops_per_edge() + (2 * s.hid * sum(s.g_cols) if k.startswith("ddp_conv_rows") else 0)) * n
                       for (s, fs), n, k in zip(self.specs, ne, self.kernel)]
            self.boundary = [n * (4.0 * fs.f_in + 32.0) + nb for (s, fs), n, nb in zip(self.specs, ne, self.node_bytes)]
            self._resolved = True

    # -- summaries ------------------------------------------------------------------------------------------------
    def hbm_summary(self, name):
        """(launches, algorithmic bytes, ms) of an HBM-bound kernel (ddp_stage_a_mfma_kernel, ddp_segment_reduce_kernel)."""
        rec = self.hbm.get(name, [])
        torch.cuda.synchronize()
        return len(rec), float(sum((r[2]() if callable(r[2]) else r[2]) for r in rec)), float(sum(r[0].elapsed_time(r[1]) for r in rec))

    def summary(self, kernel=None):
        """(launches, algorithmic FLOPs, ms) over all launches or over those of one kernel instantiation
        ("ddp_conv32_kernel": factorised shapes, "ddp_conv_messages_kernel": direct shapes)."""
        self._resolve()
        sel = [i for i, k in enumerate(self.kernel) if kernel is None or k == kernel]
        ms = sum(self.events[i][0].elapsed_time(self.events[i][1]) for i in sel)
        return len(sel), float(sum(self.flops[i] for i in sel)), float(ms)

    def split_flops(self, kernel=None, tag=False):
        """(fc-product FLOPs of the launches that ran the h2 form, all other useful FLOPs) of one kernel instantiation: the first run
        on the fp16 matrix cores at three instruction FLOPs per product FLOP, the rest (fp32-MFMA launches, G pass, contraction) in
        fp32."""
        self._resolve()
        fc16 = sum(f for f, k, h in zip(self.fc, self.kernel, self.h2) if h and (kernel is None or k == kernel))
        useful = sum(u for u, k in zip(self.useful, self.kernel) if kernel is None or k == kernel)
        return float(fc16), float(useful - fc16)

    def by_tag(self, kernel):
        """{tag: (launches, useful FLOPs, ms, edges, fc FLOPs run in the h2 form)} of one kernel instantiation, e.g. per conv layer."""
        self._resolve()
        out = {}
        for i, k in enumerate(self.kernel):
            if k != kernel:
                continue
            n, u, ms, ne, f16 = out.get(self.tags[i], (0, 0.0, 0.0, 0, 0.0))
            out[self.tags[i]] = (n + 1, u + self.useful[i], ms + self.events[i][0].elapsed_time(self.events[i][1]), ne + self.edges[i],
                                 f16 + (self.fc[i] if self.h2[i] else 0.0))
        return out

    def executed_flops(self, kernel=None):
        """FLOPs of the padded MFMA tiles + the G pass of factorised convs (a model of what is issued; the PMC pass counts it)."""
        self._resolve()
        return float(sum(e for e, k in zip(self.executed, self.kernel) if kernel is None or k == kernel))

    def useful_flops(self, kernel=None):
        """Useful fp32 FLOPs of the executed formulation without padding (packing.ConvSpec.useful_flops_per_edge)."""
        self._resolve()
        return float(sum(e for e, k in zip(self.useful, self.kernel) if kernel is None or k == kernel))

    def boundary_bytes(self):
        """Algorithmic bytes at the module boundary of the recorded conv calls (SURVEY section 8(d):
        4 (N_in D_in + E F + 4 E + N_out D_out) + 16 E per TensorProductConvLayer.forward call)."""
        self._resolve()
        return float(sum(self.boundary))


_PROFILER: Optional[ConvProfiler] = None


def set_conv_profiler(p: Optional[ConvProfiler]):
    global _PROFILER
    _PROFILER = p


def profiler(hbm=False) -> Optional[ConvProfiler]:
    p = _PROFILER
    if p is not None and hbm and not p.hbm_on:
        return None
    return p


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


# ------------------------------------------------------------------------------------------------ conv / reduce
# The fc products of the conv kernels on the fp16 matrix cores with both operands split in two halves (three products per 16 k,
# fp32 accumulation; csrc/ddp_conv.hip).  False: the exact fp32 MFMA chains of rounds 1 - 3 (A/B runs, the h2-vs-fp32 tests).
CONV_H2 = True
# where the h2 kernels report a value outside the fp16 range: an int32 in pinned host memory (score_model.overflow_flag(dev)[1:2]),
# set by the forward that is being queued (engine.forward); None: not reported
_RANGE_FLAG = None


def set_range_flag(t):
    global _RANGE_FLAG
    _RANGE_FLAG = t


# Factorised convs of the size classes ns = 60 / 32 through the 128-edge row-stationary kernel (ddp_conv_rows, csrc/ddp_conv_rows.hip; needs
# CONV_H2).  False: the 32-edge kernel of rounds 2 - 4 (A/B runs).  The two read G in different layouts: a task carries one of them.
CONV_ROWS = os.environ.get("DDP_CONV_ROWS", "1") != "0"     # (environment: same-box A/B runs)


def occupancy_shaping(rows_min_lds: int = 0, stage_a_pad: int = 0):
    """ddp_set_occupancy_shaping (include/ddp_hip.h): launches enqueued from here on - ddp_conv_rows with at least `rows_min_lds` bytes of
    dynamic LDS (> 80 KiB: one workgroup per CU), stage A's plane form with `stage_a_pad` extra bytes.  (0, 0): the kernels' own."""
    L.check(L.load().ddp_set_occupancy_shaping(int(rows_min_lds), int(stage_a_pad)), "ddp_set_occupancy_shaping")


def rows_mode(pk) -> bool:
    """Does a factorised conv with these packed weights run through ddp_conv_rows?  (Decided where stage A is planned: it writes G in
    the layout the conv kernel of the same layer reads.)"""
    return bool(CONV_H2 and CONV_ROWS and getattr(pk, "wsh", None) is not None)


def make_task(pk, x_src, ldx_src, view: EdgeView, sh, segs, msg, g=None, rows=False) -> L.ConvTask:
    """segs: [(tensor, idx_int32[E], ld, ncols)], concatenated into edge_attr_ in this order.  rows: a task of ddp_conv_rows - `g` then
    holds the G arrays in plane form (ddp_stage_a_gh)."""
    t = L.ConvTask()
    t.x_src, t.ldx_src, t.n_edges = x_src.data_ptr(), ldx_src, view.n_edges
    t.src, t.eid, t.sh = view.src.data_ptr(), view.eid.data_ptr(), sh.data_ptr()
    for k in range(L.DDP_MAX_SEGS):
        if k < len(segs):
            ten, idx, ld, n = segs[k]
            t.seg_ptr[k], t.seg_idx[k], t.seg_ld[k], t.seg_n[k] = ten.data_ptr(), idx.data_ptr(), ld, n
        else:
            t.seg_ptr[k], t.seg_idx[k], t.seg_ld[k], t.seg_n[k] = 0, 0, 0, 0
    t.w1p, t.b1p, t.w2p, t.b2p = pk.w1p.data_ptr(), pk.b1p.data_ptr(), pk.w2p.data_ptr(), pk.b2p.data_ptr()
    # fp16 hi/lo operand planes of the same weights (None / CONV_H2 off: the exact fp32 MFMA form)
    use_h2 = CONV_H2 and getattr(pk, "w1h", None) is not None and getattr(pk, "w2h", None) is not None
    t.w1h, t.w2h = (pk.w1h.data_ptr(), pk.w2h.data_ptr()) if use_h2 else (0, 0)
    t.h2_range_flag = _p(_RANGE_FLAG) if use_h2 else 0
    t.msg = msg.data_ptr()
    t.wsh, t.bsp = (pk.wsh.data_ptr(), pk.bsp.data_ptr()) if rows else (0, 0)
    for k in range(2):
        gk = g[k].data_ptr() if (g is not None and g[k] is not None) else 0
        t.g[k], t.gh[k] = (0, gk) if rows else (gk, 0)
    t.gh_fmt = int(getattr(pk, "gh_fmt", 0)) if rows else 0
    t.rows_form = int(getattr(pk, "rows_form", 0)) if rows else 0
    t.rows_bias_k = int(getattr(pk, "rows_bias_k", 0)) if rows else 0
    t.rows_seg0, t.rows_seg1 = (int(v) for v in getattr(pk, "rows_seg", (0, 0))) if rows else (0, 0)
    t.rows_nts = int(getattr(pk, "rows_nts", 0)) if rows else 0
    t._rows = bool(rows)
    t.pos = _p(view.pos)
    t.n_edges_dev = _p(view.cnt)
    t._count = (view.n_edges, view.cnt)      # (python-side only: for the profiler)
    return t


def launch_convs(spec: P.ConvSpec, tasks: List[L.ConvTask], flops_spec: Optional[P.ConvSpec] = None, node_bytes: float = 0.0, tag=None):
    """node_bytes: 4 (N_in D_in + N_out D_out) summed over the launch's conv calls; tag: position in the forward (both only used
    by the profiler)."""
    lib = L.load()
    if not tasks:
        return
    arr = (L.ConvTask * len(tasks))(*tasks)
    shape = spec.ctypes_shape()
    prof = _PROFILER
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rows = all(getattr(t, "_rows", False) for t in tasks)
    if rows:
        L.check(lib.ddp_conv_rows(C.byref(shape), arr, len(tasks), stream()), "ddp_conv_rows")
    else:
        L.check(lib.ddp_conv_messages(C.byref(shape), arr, len(tasks), stream()), "ddp_conv_messages")
    if prof is not None:
        e1.record()
        h2 = P.h2_steps(spec) > 0 and all(t.w1h and t.w2h for t in tasks)
        prof.record_conv(e0, e1, spec, flops_spec, [t._count for t in tasks], node_bytes, tag, h2,
                         rows=(2 if all(t.rows_form == 1 for t in tasks) else 1) if rows else 0)


def launch_reduce(x, ldx, n_nodes, d_out, sources, accumulate=True, n_rep=1, rep_stride=0):
    """sources: [(msg, view, packed[, rowmap])] in the reference's summation order; rowmap (int32 per CSR position, optional)
    = the row of `msg` that holds the position's message.  n_rep > 1: see ddp_segment_reduce."""
    lib = L.load()
    arr = (L.ReduceSrc * max(len(sources), 1))()
    for i, src_ in enumerate(sources):
        msg, view, pk = src_[:3]
        arr[i].msg, arr[i].rowptr = msg.data_ptr(), view.rowptr.data_ptr()
        arr[i].bn_scale, arr[i].bn_shift, arr[i].n_edges = pk.bn_scale.data_ptr(), pk.bn_shift.data_ptr(), view.n_edges
        arr[i].rowmap = src_[3].data_ptr() if (len(src_) > 3 and src_[3] is not None) else 0
        arr[i].n_edges_dev = _p(view.cnt)
    prof = profiler(hbm=True)
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(lib.ddp_segment_reduce(ptr(x), ldx, n_nodes, d_out, arr, len(sources), 1 if accumulate else 0, n_rep, rep_stride,
                                   stream()), "ddp_segment_reduce")
    if prof is not None:
        e1.record()
        # algorithmic bytes: every message row read once, every node row read and written once
        views = [s_[1] for s_ in sources]

        def nbytes(views=views, d_out=d_out, n_nodes=n_nodes, n_rep=n_rep):
            ne = sum(v.n_edges if v.cnt is None else min(v.n_edges, int(v.cnt.item())) for v in views)
            return 4.0 * d_out * (ne + 2 * n_nodes * max(n_rep, 1))
        # (the same predicate as csrc/ddp_misc.hip: the 16-byte form needs 16-byte aligned arrays and d_out, ldx multiples of 4)
        wide = (d_out % 4 == 0 and ldx % 4 == 0 and x.data_ptr() % 16 == 0 and not os.environ.get("DDP_REDUCE_NARROW")
                and all(s_[0].data_ptr() % 16 == 0 and s_[2].bn_scale.data_ptr() % 16 == 0 and s_[2].bn_shift.data_ptr() % 16 == 0 for s_ in sources))
        prof.hbm.setdefault("ddp_segment_reduce4_kernel" if wide else "ddp_segment_reduce_kernel", []).append((e0, e1, nbytes))


class EdgeMLPPack:
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


def edge_featurize(pack: EdgeMLPPack, dist, pos_a, ia, pos_b, ib, pre, pre_idx, pre2=None, n_edges=None, cnt=None):
    """pre: [*, >= ns] rows with unit column stride (a column slice of a wider table is fine); pre2 (optional, [n2, ns]) is
    added to the first n2 edges' rows (the bond-type columns of lig_edge_embedding's first Linear).  n_edges / cnt: capacity
    and device-side count of the edge arrays (default: their length, host-known)."""
    lib = L.load()
    E = int(ia.shape[0]) if n_edges is None else int(n_edges)
    dev = pos_a.device
    out = torch.empty((E, pack.ns), device=dev, dtype=torch.float32)
    sh = torch.empty((E, 4), device=dev, dtype=torch.float32)
    if E == 0:
        return out, sh
    if pre.stride(1) != 1:
        pre = pre.contiguous()
    n2 = 0 if pre2 is None else int(pre2.shape[0])
    L.check(lib.ddp_edge_featurize(ptr(pos_a), ptr(ia), ptr(pos_b), ptr(ib), E, ptr(cnt), ptr(dist.offset), pack.k,
                                   C.c_float(dist.coeff), ptr(pre), ptr(pre_idx), pre.stride(0),
                                   ptr(pre2) if n2 else None, n2, pre2.stride(0) if n2 else 0, ptr(pack.w1d),
                                   ptr(pack.w2), ptr(pack.b2), pack.ns, ptr(out), ptr(sh), stream()),
            "ddp_edge_featurize")
    return out, sh


def edge_featurize_jobs(calls):
    """Several edge_featurize calls as ONE launch (ddp_edge_featurize_jobs).  calls: [(args, kwargs)] of `edge_featurize`;
    returns [(out, sh)].  Falls back to one launch per call when an MLP is outside the batched kernel's shape range."""
    if any(a[0].k % 8 or not 8 <= a[0].k <= 64 for a, _ in calls) or len(calls) > L.DDP_MAX_FEATURIZE_JOBS:
        return [edge_featurize(*a, **kw) for a, kw in calls]
    jobs, outs, keep = [], [], []
    for (pack, dist, pos_a, ia, pos_b, ib, pre, pre_idx), kw in calls:
        pre2, n_edges, cnt = kw.get("pre2"), kw.get("n_edges"), kw.get("cnt")
        E = int(ia.shape[0]) if n_edges is None else int(n_edges)
        out = torch.empty((E, pack.ns), device=pos_a.device, dtype=torch.float32)
        sh = torch.empty((E, 4), device=pos_a.device, dtype=torch.float32)
        outs.append((out, sh))
        if E == 0:
            continue
        if pre.stride(1) != 1:
            pre = pre.contiguous()
        n2 = 0 if pre2 is None else int(pre2.shape[0])
        j = L.FeaturizeJob()
        j.pos_a, j.ia, j.pos_b, j.ib, j.n_edges, j.n_edges_dev = _p(pos_a), _p(ia), _p(pos_b), _p(ib), E, _p(cnt)
        j.offset, j.k_rbf, j.coeff = _p(dist.offset), pack.k, float(dist.coeff)
        j.pre, j.pre_idx, j.ld_pre = _p(pre), _p(pre_idx), pre.stride(0)
        j.pre2, j.n_pre2, j.ld_pre2 = (_p(pre2) if n2 else 0), n2, (pre2.stride(0) if n2 else 0)
        j.w1d, j.w2, j.b2, j.ns, j.out, j.sh = _p(pack.w1d), _p(pack.w2), _p(pack.b2), pack.ns, _p(out), _p(sh)
        jobs.append(j)
        keep.append((pre, pre2, ia, ib, pre_idx))
    if jobs:
        arr = (L.FeaturizeJob * len(jobs))(*jobs)
        L.check(L.load().ddp_edge_featurize_jobs(arr, len(jobs), stream()), "ddp_edge_featurize_jobs")
    return outs


def stage_a(x, n_rows, offs, nb, W, out, rows=None, rows_cnt=None, out_rows=None, W3=None, Wh=None, gh=None, gh_fmt=0, ldo=None):
    """ddp_stage_a: out[b][row] = x[row, offs[b]:offs[b]+k] @ W[b] for the listed rows (all n_rows rows if rows is None).
    Wh: the weights pre-split for the fp16 hi/lo form (packing.split_h2; ddp_stage_a_h2), W3: for the bf16x3 form
    (packing.split_bf16x3); neither: exact fp32 MFMA.  gh = the destination table of the plane form (int32 device tensor
    [nb, ncols / 8, 2], packing.gh_dest_table): the output leaves in the layout ddp_conv_rows reads (ddp_stage_a_gh; needs Wh)."""
    lib = L.load()
    n_in, ncols = W.shape[1], W.shape[2]
    if n_rows == 0:
        return
    if gh is not None and gh_fmt == 1:     # plane form 1 (fp16 hi + continuation bytes): rows of `ldo` floats (packing.gh3_ld = 6 ncols / 8)
        L.check(lib.ddp_stage_a_gh3(x.data_ptr(), x.stride(0), n_rows, ptr(rows), ptr(rows_cnt), out_rows if out_rows is not None else n_rows,
                                    offs, nb, W.data_ptr(), ptr(Wh), n_in, ncols, out.data_ptr(), int(ldo), ptr(_RANGE_FLAG), gh.data_ptr(),
                                    stream()), "ddp_stage_a_gh3")
        return
    if gh is not None:
        L.check(lib.ddp_stage_a_gh(x.data_ptr(), x.stride(0), n_rows, ptr(rows), ptr(rows_cnt), out_rows if out_rows is not None else n_rows,
                                   offs, nb, W.data_ptr(), ptr(Wh), n_in, ncols, out.data_ptr(), ncols, ptr(_RANGE_FLAG), gh.data_ptr(),
                                   stream()), "ddp_stage_a_gh")
        return
    if Wh is not None:
        L.check(lib.ddp_stage_a_h2(x.data_ptr(), x.stride(0), n_rows, ptr(rows), ptr(rows_cnt), out_rows if out_rows is not None else n_rows,
                                   offs, nb, W.data_ptr(), ptr(Wh), n_in, ncols, out.data_ptr(), ncols, ptr(_RANGE_FLAG), stream()), "ddp_stage_a_h2")
        return
    L.check(lib.ddp_stage_a(x.data_ptr(), x.stride(0), n_rows, ptr(rows), ptr(rows_cnt), out_rows if out_rows is not None else n_rows,
                            offs, nb, W.data_ptr(), ptr(W3), n_in, ncols, out.data_ptr(), ncols, stream()), "ddp_stage_a")


# ------------------------------------------------------------------------------------------------ list primitives
def _hold(j, *objs):
    """The job keeps the tensors whose addresses it carries alive (temporaries handed straight to a job builder would
    otherwise be freed - and their memory handed to the next temporary - before the launch)."""
    j._keep = objs
    return j


def _run_jobs(fn, cls, jobs, what):
    lib = L.load()
    for i in range(0, len(jobs), L.DDP_MAX_LIST_JOBS):
        part = jobs[i:i + L.DDP_MAX_LIST_JOBS]
        arr = (cls * len(part))(*part)
        L.check(getattr(lib, fn)(arr, len(part), stream()), what)


def scan_job(n, flag=None, val=None, rowptr=None, base=0, excl=None, excl2=None, lst=None, total=None, n_dev=None) -> L.ScanJob:
    j = L.ScanJob()
    j.n, j.n_dev, j.flag, j.val, j.rowptr, j.base = n, _p(n_dev), _p(flag), _p(val), _p(rowptr), base
    j.excl, j.excl2, j.list, j.total = _p(excl), _p(excl2), _p(lst), _p(total)
    return _hold(j, flag, val, rowptr, excl, excl2, lst, total, n_dev)


def scan_jobs(jobs: Sequence[L.ScanJob]):
    _run_jobs("ddp_scan_jobs", L.ScanJob, list(jobs), "ddp_scan_jobs")


def mark_job(mask, idx, n, n_dev=None) -> L.MarkJob:
    j = L.MarkJob()
    j.idx, j.n, j.n_dev, j.mask = _p(idx), n, _p(n_dev), _p(mask)
    return _hold(j, idx, n_dev, mask)


def mark_jobs(jobs: Sequence[L.MarkJob]):
    _run_jobs("ddp_mark_jobs", L.MarkJob, list(jobs), "ddp_mark_jobs")


def rowcopy_job(n_rows, keep, old_rowptr, new_rowptr, ins, outs) -> L.RowcopyJob:
    j = L.RowcopyJob()
    j.n_rows, j.keep, j.old_rowptr, j.new_rowptr = n_rows, _p(keep), _p(old_rowptr), _p(new_rowptr)
    for k, (a, b) in enumerate(zip(ins, outs)):
        j.inp[k], j.out[k] = _p(a), _p(b)
    return _hold(j, keep, old_rowptr, new_rowptr, list(ins), list(outs))


def rowcopy_jobs(jobs: Sequence[L.RowcopyJob]):
    _run_jobs("ddp_rowcopy_jobs", L.RowcopyJob, list(jobs), "ddp_rowcopy_jobs")


def select_job(n, mask_a, idx_a, mask_b, idx_b, pays, outs, total, scratch, out_idx=None, n_dev=None, pay_add=None) -> L.SelectJob:
    """scratch: int32 [2 * ((n + 2047) // 2048) + 1]."""
    j = L.SelectJob()
    nb = (n + 2047) // 2048
    j.n, j.n_dev, j.mask_a, j.idx_a, j.mask_b, j.idx_b, j.out_idx = n, _p(n_dev), _p(mask_a), _p(idx_a), _p(mask_b), _p(idx_b), _p(out_idx)
    for k, (a, b) in enumerate(zip(pays, outs)):
        j.pay[k], j.out[k] = _p(a), _p(b)
        j.pay_add[k] = 0 if pay_add is None else pay_add[k]
    j.total = _p(total)
    j.block_count, j.block_off = scratch.data_ptr(), scratch.data_ptr() + 4 * nb
    return _hold(j, n_dev, mask_a, idx_a, mask_b, idx_b, out_idx, list(pays), list(outs), total, scratch)


def select_jobs(jobs: Sequence[L.SelectJob]):
    _run_jobs("ddp_select_jobs", L.SelectJob, list(jobs), "ddp_select_jobs")


def radius_job(x, x_ptr, y, y_batch, r, cap, flags, counts, offsets=None, base=0, total=None, out_query=None, out_x=None,
               capacity=0, graph_div=None, overflow=None) -> L.RadiusJob:
    j = L.RadiusJob()
    j.x, j.x_ptr, j.y, j.y_batch, j.ny = x.data_ptr(), x_ptr.data_ptr(), y.data_ptr(), y_batch.data_ptr(), int(y.shape[0])
    j.r, j.max_neighbors, j.flags, j.graph_div = float(r), int(cap), int(flags), _p(graph_div)
    j.counts, j.offsets, j.base, j.total = _p(counts), _p(offsets), int(base), _p(total)
    j.out_query, j.out_x, j.capacity, j.overflow = _p(out_query), _p(out_x), int(capacity), _p(overflow)
    return _hold(j, x, x_ptr, y, y_batch, graph_div, counts, offsets, total, out_query, out_x, overflow)


def radius_search_jobs(jobs: Sequence[L.RadiusJob]):
    _run_jobs("ddp_radius_search_jobs", L.RadiusJob, list(jobs), "ddp_radius_search_jobs")


def group_job(key, n_items, n_keys, pays, rowptr, perm=None, out_key=None, outs=(), scratch=None, n_dev=None, key_map=None) -> L.GroupJob:
    """scratch: int32 [n_keys + n_items]."""
    j = L.GroupJob()
    j.key, j.n_items, j.n_items_dev, j.n_keys = _p(key), n_items, _p(n_dev), n_keys
    for k, a in enumerate(pays):
        j.pay[k] = _p(a)
    j.rowptr, j.perm, j.out_key, j.key_map, j.scratch = _p(rowptr), _p(perm), _p(out_key), _p(key_map), _p(scratch)
    for k, a in enumerate(outs):
        j.out[k] = _p(a)
    return _hold(j, key, n_dev, list(pays), rowptr, perm, out_key, list(outs), key_map, scratch)


def group_jobs(jobs: Sequence[L.GroupJob]):
    _run_jobs("ddp_group_by_key_jobs", L.GroupJob, list(jobs), "ddp_group_by_key_jobs")


def gather_rows(x, idx, n, out, ncols, n_dev=None):
    lib = L.load()
    if n > 0:
        L.check(lib.ddp_gather_rows(x.data_ptr(), x.stride(0), idx.data_ptr(), n, ptr(n_dev), out.data_ptr(), out.stride(0), ncols,
                                    stream()), "ddp_gather_rows")


def flex_mark(a, b, n_edges, e0, n_a_per_graph, n_b_per_graph, mark, pos=None, flag=None, a_too=False, ref_list=False):
    """ddp_flex_mark: mark[a[e]] = 1 where b[e] (a_too: or a[e]) is "off" in its sample - by position (pos) or by an earlier mask (flag)."""
    L.check(L.load().ddp_flex_mark(ptr(pos), ptr(flag), n_b_per_graph, ptr(a), ptr(b), n_edges, e0, n_a_per_graph, int(a_too), int(ref_list),
                                   ptr(mark), stream()), "ddp_flex_mark")


def fallback_rowmap(mark, recv, old_rowptr, new_rowptr, n_edges, n_recv_per_graph, rowmap):
    L.check(L.load().ddp_fallback_rowmap(ptr(mark), ptr(recv), ptr(old_rowptr), ptr(new_rowptr), n_edges, n_recv_per_graph, ptr(rowmap),
                                         stream()), "ddp_fallback_rowmap")


def clean_pair_maps(touched, recv, src, n_edges, e0, n_graphs, n0, rowmap, rows_v, rowptr=None):
    lib = L.load()
    L.check(lib.ddp_clean_pair_maps(touched.data_ptr(), recv.data_ptr(), src.data_ptr(), ptr(rowptr), n_edges, e0, n_graphs, n0,
                                    rowmap.data_ptr(), rows_v.data_ptr(), stream()), "ddp_clean_pair_maps")
