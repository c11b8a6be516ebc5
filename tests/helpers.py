"""Shared test helpers: load a golden case, rebuild its seeded weights and inputs."""
import os

import torch

from oracle.cases import CASES, input_checksums
from oracle.weights import synth_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, f"{name}.pt"), weights_only=False)


def golden_state_dict(gold, seed):
    template = {k: tuple(v) for k, v in gold["state_dict_shapes"].items()}
    template.update(gold["offsets"])
    return synth_state_dict(template, seed)


def case_inputs(name):
    case = CASES[name]
    gold = load_golden(name)
    batch = case.make_batch()
    chk = input_checksums(batch)
    for k, v in gold["inputs"].items():
        assert abs(chk[k] - v) <= 1e-6 * max(1.0, abs(v)), f"regenerated input {k} differs from the golden run"
    return case, gold, batch, golden_state_dict(gold, case.weight_seed)


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): relative to the largest reference component (fp32 parity metric)."""
    if a.numel() == 0 and b.numel() == 0:
        return 0.0
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def elementwise_excess(a, b, rtol=1e-4, atol_frac=1e-6):
    """Element-wise parity metric for per-bond score arrays (tor / sc_tor), where one large bond score must not hide
    errors on the small ones: max over elements of |a-b| / (rtol*|b| + atol_frac*max|b|); <= 1 passes."""
    if a.numel() == 0 and b.numel() == 0:
        return 0.0
    assert a.shape == b.shape, (a.shape, b.shape)
    a, b = a.double(), b.double()
    bound = rtol * b.abs() + atol_frac * b.abs().max().clamp_min(1e-30)
    return float(((a - b).abs() / bound).max())


def rowwise_excess(a, b, rtol=1e-4, atol_frac=1e-6):
    """Per-row parity metric for the [B, 3] tr / rot score arrays: one graph with a large score must not hide an error on a
    graph with a small one.  max over rows of |a_row - b_row|_inf / (rtol * |b_row|_inf + atol_frac * max|b|); <= 1 passes."""
    if a.numel() == 0 and b.numel() == 0:
        return 0.0
    assert a.shape == b.shape, (a.shape, b.shape)
    a, b = a.double().reshape(a.shape[0], -1), b.double().reshape(b.shape[0], -1)
    bound = rtol * b.abs().amax(1) + atol_frac * b.abs().max().clamp_min(1e-30)
    return float(((a - b).abs().amax(1) / bound).max())


def conv_stats_excess(got, stats, rtol=1e-4, atol_frac=1e-4):
    """Compares one conv output [n, d] of the HIP model with the golden `conv_stats` entry the reference's forward hook left
    (oracle/make_golden.py: shape, mean|.|, a 64-point strided sample of the flattened output).  Returns the worst ratio of
    |sample difference| to (rtol * |ref| + atol_frac * mean|ref|) and the relative difference of mean|.|."""
    import torch
    assert list(got.shape) == list(stats["shape"]), (tuple(got.shape), stats["shape"])
    flat = got.reshape(-1).double().cpu()
    ref = stats["sample"].double()
    if flat.numel() > ref.numel():
        flat = flat[torch.linspace(0, flat.numel() - 1, ref.numel()).long()]
    bound = rtol * ref.abs() + atol_frac * max(stats["mean_abs"], 1e-30)
    worst = float(((flat - ref).abs() / bound).max()) if ref.numel() else 0.0
    mean_rel = abs(float(got.abs().mean()) - stats["mean_abs"]) / max(stats["mean_abs"], 1e-30) if got.numel() else 0.0
    return worst, mean_rel


class PyGLikeStore:
    """Storage of PyGLikeBatch: attributes live in a mapping; `len` = number of attributes, `in` by attribute name."""

    def __init__(self):
        object.__setattr__(self, "_mapping", {})

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_mapping")[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self._mapping[k] = v

    def __getitem__(self, k):
        return self._mapping[k]

    def __setitem__(self, k, v):
        self._mapping[k] = v

    def __contains__(self, k):
        return k in self._mapping

    def __len__(self):
        return len(self._mapping)


class PyGLikeBatch:
    """An object with the ACCESS SEMANTICS of torch_geometric 2.4's HeteroData batch (restated from its documented behaviour;
    torch_geometric itself is not installed): node stores under str keys, edge stores under canonical 3-tuples
    (src, rel, dst) that also answer to (src, dst); a store is created on first access; `key in data` looks at ATTRIBUTE names
    (BaseData.__contains__), not at node types; global attributes (complex_t, num_graphs) by attribute access.  Shares no
    code with diffdock_pocket_amd.batch.HeteroBatch: the drop-in must only rely on what a PyG batch offers."""

    def __init__(self):
        object.__setattr__(self, "_node", {})
        object.__setattr__(self, "_edge", {})
        object.__setattr__(self, "_glob", PyGLikeStore())

    def _edge_key(self, key):
        if len(key) == 3:
            return tuple(key)
        hits = [k for k in self._edge if k[0] == key[0] and k[2] == key[1]]
        return hits[0] if hits else (key[0], "to", key[1])

    def __getitem__(self, key):
        if isinstance(key, tuple):
            return self._edge.setdefault(self._edge_key(key), PyGLikeStore())
        return self._node.setdefault(key, PyGLikeStore())

    def __delitem__(self, key):
        if isinstance(key, tuple):
            del self._edge[self._edge_key(key)]
        else:
            del self._node[key]

    def __contains__(self, key):      # attribute names over all stores, as PyG does
        return any(key in st for st in [self._glob, *self._node.values(), *self._edge.values()])

    def __getattr__(self, k):
        return getattr(object.__getattribute__(self, "_glob"), k)

    def __setattr__(self, k, v):
        setattr(self._glob, k, v)

    @classmethod
    def from_hetero_batch(cls, b, device, rel=None):
        """Copy of a collated diffdock_pocket_amd HeteroBatch (test inputs are generated as such) with tensors on `device`."""
        import torch
        rel = rel or {("ligand", "ligand"): "lig_bond", ("receptor", "receptor"): "rec_contact", ("atom", "atom"): "atom_contact",
                      ("atom", "receptor"): "atom_rec_contact"}

        def mv(v):
            if torch.is_tensor(v):
                return v.to(device)
            if isinstance(v, dict):
                return {k: mv(x) for k, x in v.items()}
            return v

        out = cls()
        for key, st in b._stores.items():
            dst = out[(key[0], rel.get(key, "to"), key[1])] if isinstance(key, tuple) else out[key]
            for a, v in st.items():
                setattr(dst, a, mv(v))
        for a, v in b._globals.items():
            setattr(out, a, mv(v))
        return out


def decode_gh_rows(rows, widths, hid, fmt):
    """Decodes G rows in the plane forms of ddp_conv_task_t::gh (include/ddp_hip.h) back to fp64: rows [N, ld] float32 (the bytes stage A
    wrote) -> (V [N, n8, gcp, 8] = the plane values hi + lo per k8 group, padded column and k slot; Gb [N, gcp]).  fmt 0: hi / lo fp16 words
    side by side; fmt 1: 24-byte units of 8 hi words (V truncated) and 8 continuation bytes (19 significant bits).  Test infrastructure: written from
    the header's description of the layouts, independently of packing.gh_dest_table."""
    import numpy as np
    import torch
    n8, gcp = (hid + 7) // 8, sum(widths)
    raw = rows.detach().cpu().contiguous().numpy().view(np.uint8)          # [N, 4 ld] bytes
    N = raw.shape[0]
    V = np.zeros((N, n8, gcp, 8), dtype=np.float64)
    Gb = np.zeros((N, gcp), dtype=np.float64)

    cum = 0
    for w in widths:
        if fmt == 0:
            base = 2 * n8 * cum * 16
            t = raw[:, base:base + n8 * w * 32].reshape(N, n8, w, 2, 16)
            hi = t[:, :, :, 0].copy().view(np.float16).astype(np.float64)
            lo = t[:, :, :, 1].copy().view(np.float16).astype(np.float64)
        else:
            # plane form 1 (ABI 17): unit (k8, c) = 8 fp16 hi words + 8 continuation bytes; V = hi + sign(hi) 2^E(hi) u8 / 2^18, zero exponent: hi
            base = n8 * cum * 24
            t = raw[:, base:base + n8 * w * 24].reshape(N, n8, w, 24)
            hw = t[..., :16].copy().view(np.uint16)
            hi = hw.view(np.float16).astype(np.float64)
            e = ((hw >> 10) & 31).astype(np.int64)
            mag = np.where(e > 0, np.ldexp(t[..., 16:].astype(np.float64), e - 15 - 18), 0.0)
            lo = np.where((hw >> 15) == 1, -mag, mag)
        V[:, :, cum:cum + w] = hi + lo
        cum += w
    f32 = raw.view(np.float32)
    if fmt != 1:
        Gb[:] = f32[:, 8 * n8 * gcp:8 * n8 * gcp + gcp]
    else:
        g0 = 24 * n8 * gcp // 4
        c = np.arange(gcp)
        Gb[:] = f32[:, g0 + 6 * (c // 6) + c % 6]
    return torch.from_numpy(V), torch.from_numpy(Gb)
