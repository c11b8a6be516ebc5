"""Graph construction for the score-model forward.

The counterparts of the torch_cluster calls inside the reference forward (models/all_atom_score_model.py:457,524,
545-564,607,627) with the conventions restated in SURVEY Appendix B.3.  For tensors on the ROCm device the searches run
in csrc/ddp_graph.hip (ddp_radius_count / ddp_radius_fill / ddp_knn: one thread per query scanning its graph's points,
no dense distance blocks); the dense PyTorch formulation below is their definition - it is what the CPU tests pin to
the oracle's torch_cluster restatement, and what the HIP kernels are tested against bit for bit on the GPU.

  radius(x, y, r, batch_x, batch_y, max_num_neighbors) -> [2,E]: row0 = query (y) index, row1 = x index,
      strict '<', same graph only, at most max_num_neighbors per query, emitted query-major with ascending x index.
      Truncation rule (`TRUNCATION`, or the `truncation=` argument): "first_index" (default) keeps the first
      max_num_neighbors matches in ascending x index - what torch_cluster's CUDA kernel, i.e. the reference on a GPU,
      does; "nearest" keeps the nearest ones (and ties).  torch_cluster's CPU path (nanoflann, unsorted) is neither.
  radius_graph(x, r, batch)      -> [neighbour; query], self loops dropped, cap 32 (+1 internally).
  knn_graph(x, k, batch)         -> [neighbour; query], k nearest by distance, self excluded.

Implementation: graphs in a batch are padded to a dense [B, n_max] layout so distances are one batched
`cdist`-like computation per edge type ([B, ny, nx]; for 40 x 1111 atoms that is 49 M floats - sized for 288 GB of
HBM, no [sum N]^2 blow-up).  The dense layout of a node set is cached in `DenseLayout` and reused across calls
while the batch vector is the same tensor.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


TRUNCATION = "first_index"     # rule of radius() / radius_graph() when a query has more matches than its cap
_FLAG_DROP_SELF, _FLAG_NEAREST = 1, 2   # DDP_RADIUS_* of include/ddp_hip.h


def _rule(truncation):
    t = TRUNCATION if truncation is None else truncation
    if t not in ("first_index", "nearest"):
        raise ValueError(f"unknown radius truncation rule {t!r}")
    return t


def _raw_stream():
    """Raw handle of the current HIP stream of the current device (one C call; torch.cuda.current_stream() is ~9 us of Python)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


@dataclass
class DenseLayout:
    """Padded [B, nmax] view of a sorted batch vector."""
    B: int
    nmax: int
    counts: torch.Tensor      # [B]
    starts: torch.Tensor      # [B]
    slot: torch.Tensor        # [N] position of node inside its graph
    index: torch.Tensor       # [B, nmax] global node id or -1
    uniform: bool
    counts_host: tuple = ()   # the graph sizes as python ints (capacities of the pose-dependent edge lists are sized from them)

    @staticmethod
    def build(batch: torch.Tensor, B: int) -> "DenseLayout":
        """(synchronises with the host once - the graph sizes are read back; a layout is built once per batch vector)"""
        counts = torch.bincount(batch, minlength=B)
        starts = torch.cumsum(counts, 0) - counts
        counts_host = tuple(int(c) for c in counts.tolist())
        nmax = max(counts_host) if counts_host else 0
        n = batch.shape[0]
        slot = torch.arange(n, device=batch.device) - starts[batch]
        index = torch.full((B, max(nmax, 1)), -1, dtype=torch.long, device=batch.device)
        index[batch, slot] = torch.arange(n, device=batch.device)
        uniform = all(c == nmax for c in counts_host)
        return DenseLayout(B, nmax, counts, starts, slot, index, uniform, counts_host)

    def dense(self, values: torch.Tensor, fill: float) -> torch.Tensor:
        """[N, d] -> [B, nmax, d] (padding rows = fill).  No host synchronisation (no boolean-mask indexing)."""
        if self.uniform:
            return values.reshape(self.B, self.nmax, *values.shape[1:])
        pad = self.index < 0
        out = values[self.index.clamp(min=0)]
        return out.masked_fill(pad.reshape(pad.shape + (1,) * (values.dim() - 1)), fill)


def _sqdist(y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """[B, ny, 3], [B, nx, 3] -> [B, ny, nx] squared distances, direct differences (no |x|^2+|y|^2-2xy trick, so
    the strict '<' test sees the same fp32 values as a per-pair kernel would)."""
    d = y.unsqueeze(2) - x.unsqueeze(1)
    return (d * d).sum(-1)


def _i32(t):
    return t.to(torch.int32).contiguous()


def _ptr(layout: "DenseLayout") -> torch.Tensor:
    """int32 [B+1] node offsets of the graphs of a layout (cached on the layout)."""
    p = getattr(layout, "_ptr32", None)
    if p is None:
        p = torch.zeros(layout.B + 1, dtype=torch.int32, device=layout.counts.device)
        p[1:] = torch.cumsum(layout.counts, 0).to(torch.int32)
        layout._ptr32 = p
        layout._batch32 = None
    return p


class RadiusSearch:
    """radius() / radius_graph() on the device (csrc/ddp_graph.hip) in two halves - count pass + prefix sum, then fill pass -
    so that several searches of a forward share ONE host synchronisation (`resolve`) and a search whose edges are only
    needed late (the heads' bond-centre graphs) can be counted early and filled after the conv layers are queued.

        s = RadiusSearch(x, y, r, lx, ly, cap)          # edges [query (y); x], as radius()
        s = RadiusSearch.graph(x, r, lx, cap)           # edges [neighbour; query], self loops dropped, as radius_graph()
        (E,) = resolve([s]);  edge_index = s.fill(E)
    """

    def __init__(self, x, y, r, lx, ly, max_num_neighbors=32, drop_self=False, flip=False, truncation=None):
        from . import _lib as L
        lib = L.load()
        flags = (_FLAG_DROP_SELF if drop_self else 0) | (_FLAG_NEAREST if _rule(truncation) == "nearest" else 0)
        self.args = (x.float().contiguous(), y.float().contiguous(), float(r), _ptr(lx), _batch32(ly, y.shape[0]),
                     int(max_num_neighbors), flags)
        self.flip = flip
        xc, yc, r, xptr, ybatch, cap, ds = self.args
        ny = yc.shape[0]
        counts = torch.empty(ny, dtype=torch.int32, device=xc.device)
        if ny > 0:
            L.check(lib.ddp_radius_count(xc.data_ptr(), xptr.data_ptr(), yc.data_ptr(), ybatch.data_ptr(), ny, r, cap, ds,
                                         counts.data_ptr(), _raw_stream()), "ddp_radius_count")
        self.counts = counts                     # matches per query
        self.offs = torch.zeros(ny + 1, dtype=torch.int32, device=xc.device)
        self.offs[1:] = torch.cumsum(counts, 0)

    @classmethod
    def graph(cls, x, r, lx, max_num_neighbors=32, truncation=None):
        return cls(x, x, r, lx, lx, max_num_neighbors + 1, drop_self=True, flip=True, truncation=truncation)

    def fill(self, E: int) -> torch.Tensor:
        from . import _lib as L
        lib = L.load()
        xc, yc, r, xptr, ybatch, cap, ds = self.args
        oq = torch.empty(E, dtype=torch.int32, device=xc.device)
        ox = torch.empty(E, dtype=torch.int32, device=xc.device)
        if E > 0:
            L.check(lib.ddp_radius_fill(xc.data_ptr(), xptr.data_ptr(), yc.data_ptr(), ybatch.data_ptr(), yc.shape[0], r, cap, ds,
                                        self.offs.data_ptr(), oq.data_ptr(), ox.data_ptr(),
                                        _raw_stream()), "ddp_radius_fill")
        self.row32 = (ox, oq) if self.flip else (oq, ox)     # int32 rows of the returned edge_index
        return torch.stack([ox.long(), oq.long()] if self.flip else [oq.long(), ox.long()], 0)


def resolve(searches, extra=()):
    """Edge counts of several RadiusSearch objects - and the values of `extra` 0-dim device tensors the caller wants on the
    host at the same time - with ONE device-to-host copy (the only host synchronisation of the searches).
    Returns the counts as ints followed by the extra values as ints."""
    items = [s.offs[-1] for s in searches] + [e.to(torch.int32) for e in extra]
    if not items:
        return []
    return [int(v) for v in torch.stack(items).tolist()]


def _batch32(layout: "DenseLayout", n: int) -> torch.Tensor:
    """int32 [N] graph index per node, from the layout (cached)."""
    _ptr(layout)
    if layout._batch32 is None:
        layout._batch32 = torch.repeat_interleave(torch.arange(layout.B, device=layout.counts.device, dtype=torch.int32),
                                                  layout.counts, output_size=n)
    return layout._batch32


def radius(x, y, r, lx: DenseLayout, ly: DenseLayout, max_num_neighbors=32, truncation=None):
    """See module docstring.  `r` may be a python float or a [B] tensor is NOT supported (scale inputs instead,
    as the reference does for the dynamic cross cutoff)."""
    if x.is_cuda:   # device search (no dense [B, ny, nx] blocks); the PyTorch form below is the CPU / test definition
        s = RadiusSearch(x, y, r, lx, ly, max_num_neighbors, truncation=truncation)
        return s.fill(resolve([s])[0])
    xd, yd = lx.dense(x, float("inf")), ly.dense(y, float("inf"))
    d2 = _sqdist(yd, xd)
    ok = d2 < (float(r) ** 2)
    if not (lx.uniform and ly.uniform):
        ok = ok & (ly.index >= 0).unsqueeze(2) & (lx.index >= 0).unsqueeze(1)
    if lx.nmax > max_num_neighbors:
        cnt_max = int(ok.sum(-1).max().item()) if ok.numel() else 0
        if cnt_max > max_num_neighbors and _rule(truncation) == "nearest":
            d2m = torch.where(ok, d2, torch.full_like(d2, float("inf")))
            kth = torch.topk(d2m, max_num_neighbors, dim=-1, largest=False).values[..., -1:]
            ok = ok & (d2m <= kth)
        elif cnt_max > max_num_neighbors:      # the first max_num_neighbors matches in ascending x index
            ok = ok & (torch.cumsum(ok.to(torch.int32), dim=-1) <= max_num_neighbors)
    b, q, n = ok.nonzero(as_tuple=True)
    return torch.stack([ly.index[b, q], lx.index[b, n]], 0)


def radius_graph(x, r, lx: DenseLayout, max_num_neighbors=32, truncation=None):
    if x.is_cuda:
        s = RadiusSearch.graph(x, r, lx, max_num_neighbors, truncation=truncation)
        return s.fill(resolve([s])[0])
    ei = radius(x, x, r, lx, lx, max_num_neighbors + 1, truncation=truncation)
    keep = ei[0] != ei[1]
    return torch.stack([ei[1][keep], ei[0][keep]], 0)


def knn_graph(x, k, lx: DenseLayout):
    if x.is_cuda:
        from . import _lib as L
        lib = L.load()
        n = x.shape[0]
        kk = min(k, max(lx.nmax - 1, 0))
        if kk == 0 or n == 0:
            return torch.zeros((2, 0), dtype=torch.long, device=x.device)
        xc = x.float().contiguous()
        nb = torch.empty((n, kk), dtype=torch.int32, device=x.device)
        L.check(lib.ddp_knn(xc.data_ptr(), _ptr(lx).data_ptr(), _batch32(lx, n).data_ptr(), n, kk, nb.data_ptr(),
                            _raw_stream()), "ddp_knn")
        q = torch.arange(n, device=x.device).unsqueeze(1).expand(n, kk)
        if lx.uniform or min(lx.counts_host) > kk:       # every node has kk neighbours: no compaction
            return torch.stack([nb.reshape(-1).long(), q.reshape(-1)], 0)
        keep = nb >= 0
        return torch.stack([nb[keep].long(), q[keep]], 0)
    xd = lx.dense(x, float("inf"))
    d2 = _sqdist(xd, xd)
    if not lx.uniform:
        pad = lx.index < 0
        d2 = d2.masked_fill(pad.unsqueeze(1) | pad.unsqueeze(2), float("inf"))
    d2 = torch.where(torch.isnan(d2), torch.full_like(d2, float("inf")), d2)
    ar = torch.arange(d2.shape[1], device=x.device)
    d2[:, ar, ar] = float("inf")
    kk = min(k, max(lx.nmax - 1, 0))
    if kk == 0:
        return torch.zeros((2, 0), dtype=torch.long, device=x.device)
    val, idx = torch.topk(d2, kk, dim=-1, largest=False, sorted=True)     # [B, n, kk]
    q = lx.index.unsqueeze(-1).expand_as(idx)
    nb = torch.gather(lx.index.unsqueeze(1).expand(-1, d2.shape[1], -1), 2, idx)
    keep = torch.isfinite(val) & (q >= 0)
    return torch.stack([nb[keep], q[keep]], 0)


@dataclass
class EdgeView:
    """Edges of one conv direction as the kernels consume them: CSR order of the receiving node (`rowptr` set) or source-node
    order of a factorised conv (`pos` set: the message row of every listed edge).  `n_edges` is the CAPACITY of the arrays;
    `cnt` (int32 [1] on the device, or None when the capacity IS the count) holds the actual number of edges
    (include/ddp_hip.h, "Device-side counts")."""
    n_edges: int
    recv: torch.Tensor     # int32 [E] receiving node per position
    src: torch.Tensor      # int32 [E] feature-source node per position
    eid: torch.Tensor      # int32 [E] canonical edge id (row of edge_base / edge_sh)
    rowptr: Optional[torch.Tensor] = None   # int32 [n_recv + 1] (CSR views)
    pos: Optional[torch.Tensor] = None      # int32 [E] message row (source-ordered views)
    cnt: Optional[torch.Tensor] = None

    def prefix(self, n: int, n_rows: Optional[int] = None) -> "EdgeView":
        """The first n edges (and, for a CSR view, the first n_rows receiving nodes) with a host-known count."""
        return EdgeView(n, self.recv[:n], self.src[:n], self.eid[:n],
                        self.rowptr[:n_rows + 1] if (self.rowptr is not None and n_rows is not None) else self.rowptr,
                        self.pos[:n] if self.pos is not None else None, None)


def CSR(n_edges, recv, src, eid, rowptr) -> EdgeView:
    """Edges in CSR order of the receiving node (what ddp_conv_messages / ddp_segment_reduce consume)."""
    return EdgeView(n_edges, recv, src, eid, rowptr=rowptr)


def SourceOrder(n_edges, recv, src, eid, pos) -> EdgeView:
    """The edges of a CSR view re-listed in SOURCE-node order (for the factorised conv, which streams one G[j] per
    source node): `pos` = the row of each listed edge in the receiver-CSR message array."""
    return EdgeView(n_edges, recv, src, eid, pos=pos)


def _group_by_key(key32, n_keys, pays, want_key=True, want_perm=True):
    """csrc/ddp_views.hip (ddp_group_by_key): stable grouping of int32 items on the device, see include/ddp_hip.h."""
    from . import _lib as L
    lib = L.load()
    E, dev = int(key32.shape[0]), key32.device
    i32 = dict(dtype=torch.int32, device=dev)
    rowptr = torch.empty(n_keys + 1, **i32)
    scratch = torch.empty(n_keys + E, **i32)
    perm = torch.empty(E, **i32) if want_perm else None
    out_key = torch.empty(E, **i32) if (want_perm and want_key) else None
    outs = [torch.empty(E, **i32) if want_perm else None for _ in pays]
    ptr = lambda t: t.data_ptr() if t is not None else None          # noqa: E731
    pp = [ptr(p) for p in pays] + [None] * (3 - len(pays))
    oo = [ptr(o) for o in outs] + [None] * (3 - len(outs))
    L.check(lib.ddp_group_by_key(key32.data_ptr(), E, n_keys, pp[0], pp[1], pp[2], rowptr.data_ptr(), ptr(perm), ptr(out_key),
                                 oo[0], oo[1], oo[2], scratch.data_ptr(), _raw_stream()),
            "ddp_group_by_key")
    return rowptr, perm, out_key, outs


_IOTA = {}


def iota32(n: int, dev) -> torch.Tensor:
    """int32 [0, n) on `dev` as a view of a cached, growing arange (read-only by convention; no launch per call)."""
    buf = _IOTA.get(dev)
    if buf is None or buf.shape[0] < n:
        buf = _IOTA[dev] = torch.arange(max(2 * n, 1 << 16), dtype=torch.int32, device=dev)
    return buf[:n]


def _as_i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


def build_csr(recv: torch.Tensor, src: torch.Tensor, n_recv: int, presorted: bool = False) -> EdgeView:
    """recv / src: int64 or int32 [E].  On the device this is ONE ddp_group_by_key call (5 launches); the PyTorch form below
    is its definition (CPU tests; tests/test_gpu_parity.py compares the two bit for bit)."""
    E = int(recv.shape[0])
    dev = recv.device
    if E == 0:
        z = torch.zeros(0, dtype=torch.int32, device=dev)
        return CSR(0, z, z, z, torch.zeros(n_recv + 1, dtype=torch.int32, device=dev))
    if recv.is_cuda:
        r32, s32 = _as_i32(recv).contiguous(), _as_i32(src).contiguous()
        if presorted:
            rowptr, _, _, _ = _group_by_key(r32, n_recv, [], want_perm=False)
            return CSR(E, r32, s32, iota32(E, dev), rowptr)
        rowptr, perm, r_sorted, (s_sorted,) = _group_by_key(r32, n_recv, [s32])
        return CSR(E, r_sorted, s_sorted, perm, rowptr)
    if presorted:
        perm = torch.arange(E, device=dev)
        r_sorted = recv
    else:
        r_sorted, perm = torch.sort(recv, stable=True)
    # (torch.bincount synchronises with the host to size its output; the node count is known here)
    counts = torch.zeros(n_recv, dtype=torch.int64, device=dev).index_add_(0, r_sorted.long(), torch.ones_like(r_sorted, dtype=torch.int64))
    rowptr = torch.zeros(n_recv + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return CSR(E, r_sorted.to(torch.int32), src[perm].to(torch.int32), perm.to(torch.int32), rowptr)


def source_order(csr: EdgeView, n_src: Optional[int] = None) -> EdgeView:
    """n_src (number of source nodes, an upper bound of csr.src + 1) selects the device grouping kernel; without it the
    PyTorch stable sort is used (same result)."""
    if csr.n_edges == 0:
        return SourceOrder(0, csr.recv, csr.src, csr.eid, csr.eid)
    if csr.src.is_cuda and n_src is not None:
        _, pos, src_sorted, (recv_s, eid_s) = _group_by_key(csr.src.contiguous(), int(n_src), [csr.recv.contiguous(), csr.eid.contiguous()])
        return SourceOrder(csr.n_edges, recv_s, src_sorted, eid_s, pos)
    _, order = torch.sort(csr.src.long(), stable=True)
    return SourceOrder(csr.n_edges, csr.recv[order].contiguous(), csr.src[order].contiguous(),
                       csr.eid[order].contiguous(), order.to(torch.int32).contiguous())
