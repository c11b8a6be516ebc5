"""Minimal heterogeneous-graph batch container for the score-model boundary.

The reference hands `TensorProductScoreModel.forward` a torch_geometric `HeteroDataBatch`
(reference utils/sampling.py:112-120).  torch_geometric is not a dependency of this package; `HeteroBatch`
quacks like that object for exactly the accesses the forward performs:

    data['ligand'].x / .pos / .batch / .edge_mask / .node_t[...]     node stores
    data['ligand', 'ligand'].edge_index / .edge_attr                  edge stores; a 3-tuple key
    data['ligand', 'lig_bond', 'ligand'] resolves to the same store   (PyG semantics)
    data['flexResidues'].edge_idx / .batch, len(data['flexResidues'])
    data.complex_t[...], data.num_graphs
    attribute writes: data[nt].node_sigma_emb, data['atom','atom'].edge_index, data.graph_sigma_emb

A real PyG batch can be passed to the model as well; the model only uses the accesses above.
`collate` follows PyG 2.4 `Batch.from_data_list` for the fields used: node stores are
concatenated, attributes whose key contains "index" are offset by the node counts, `batch` vectors are created,
`flexResidues.edge_idx` is NOT offset (the model adds per-graph atom offsets itself, reference
models/all_atom_score_model.py:638-652).
"""
from __future__ import annotations

import copy
from typing import Dict, Iterable, List

import torch


class Store:
    """Attribute bag (node store or edge store)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __len__(self):
        return len(self.__dict__)

    def __contains__(self, k):
        return k in self.__dict__

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    @property
    def num_nodes(self):
        if "_num_nodes" in self.__dict__:
            return self.__dict__["_num_nodes"]
        for k in ("x", "pos", "batch", "edge_idx"):
            if k in self.__dict__ and torch.is_tensor(self.__dict__[k]):
                return self.__dict__[k].shape[0]
        raise AttributeError("num_nodes")

    @num_nodes.setter
    def num_nodes(self, v):
        self.__dict__["_num_nodes"] = v

    def __repr__(self):
        return "Store(" + ", ".join(f"{k}={tuple(v.shape) if torch.is_tensor(v) else type(v).__name__}"
                                    for k, v in self.__dict__.items()) + ")"


def _apply(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _apply(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_apply(v, fn) for v in obj)
    return obj


class HeteroBatch:
    def __init__(self):
        object.__setattr__(self, "_stores", {})
        object.__setattr__(self, "_globals", {})

    # -- mapping access -------------------------------------------------------------------------
    @staticmethod
    def _norm(key):
        if isinstance(key, tuple):
            if len(key) == 3:
                key = (key[0], key[2])
            assert len(key) == 2
        return key

    def __getitem__(self, key) -> Store:
        key = self._norm(key)
        st = self._stores.get(key)
        if st is None:
            st = self._stores[key] = Store()
        return st

    def __setitem__(self, key, value):
        self._stores[self._norm(key)] = value

    def __delitem__(self, key):
        del self._stores[self._norm(key)]

    def __contains__(self, key):
        return self._norm(key) in self._stores

    # -- global attributes (complex_t, num_graphs, graph_sigma_emb, ...) -------------------------
    def __getattr__(self, name):
        g = object.__getattribute__(self, "_globals")
        if name in g:
            return g[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self._globals[name] = value

    @property
    def node_types(self):
        return [k for k in self._stores if not isinstance(k, tuple)]

    @property
    def edge_types(self):
        return [k for k in self._stores if isinstance(k, tuple)]

    def to(self, device):
        out = HeteroBatch()
        for k, st in self._stores.items():
            out._stores[k] = Store(**{a: _apply(v, lambda t: t.to(device)) for a, v in st.items()})
        for k, v in self._globals.items():
            out._globals[k] = _apply(v, lambda t: t.to(device))
        return out

    def clone(self):
        return copy.deepcopy(self)

    def __repr__(self):
        return "HeteroBatch(" + ", ".join(f"{k}: {v}" for k, v in self._stores.items()) + ")"


def collate(graphs: Iterable[HeteroBatch]) -> HeteroBatch:
    """Concatenate single-complex graphs into one batch (see module docstring)."""
    graphs = list(graphs)
    out = HeteroBatch()
    node_types = graphs[0].node_types
    offsets: Dict[str, List[int]] = {}
    for nt in node_types:
        sts = [g[nt] for g in graphs]
        if all(len(s) == 0 for s in sts):   # a store that only exists because it was looked up (no flexible residues)
            continue
        counts = [s.num_nodes for s in sts]
        off = [0]
        for c in counts:
            off.append(off[-1] + c)
        offsets[nt] = off
        new = Store()
        for a in sts[0].keys():
            if a.startswith("_"):
                continue
            v0 = getattr(sts[0], a)
            if torch.is_tensor(v0):
                setattr(new, a, torch.cat([getattr(s, a) for s in sts], 0))
            elif isinstance(v0, dict):
                setattr(new, a, {k: torch.cat([getattr(s, a)[k] for s in sts], 0) for k in v0})
            else:  # python / numpy side data (e.g. mask_rotate): keep as list, like PyG
                setattr(new, a, [getattr(s, a) for s in sts])
        dev = next((v.device for v in new.__dict__.values() if torch.is_tensor(v)), torch.device("cpu"))
        new.batch = torch.repeat_interleave(torch.arange(len(graphs), device=dev),
                                            torch.tensor(counts, device=dev))
        if nt == "flexResidues":
            new.num_nodes = off[-1]
        out[nt] = new
    for et in graphs[0].edge_types:
        src, dst = et
        new = Store()
        for a in graphs[0][et].keys():
            vs = [getattr(g[et], a) for g in graphs]
            if "index" in a:
                shift = [torch.tensor([[offsets[src][i]], [offsets[dst][i]]], device=vs[i].device, dtype=vs[i].dtype)
                         for i in range(len(graphs))]
                setattr(new, a, torch.cat([v + s for v, s in zip(vs, shift)], 1))
            else:
                setattr(new, a, torch.cat(vs, 0))
        out[et] = new
    out.num_graphs = len(graphs)
    return out


def set_time(batch: HeteroBatch, t_tr, t_rot, t_tor, t_sc_tor, device=None) -> HeteroBatch:
    """Counterpart of reference utils/diffusion_utils.py:124-165 for the all-atom case: constant per-node and
    per-graph time tensors.  One `torch.full` per distinct time value and store (the sampler passes the same t four times:
    the four keys then share one tensor, which nothing downstream writes to) instead of `v * torch.ones(n)` per key."""
    device = device or batch["ligand"].pos.device
    b = batch.num_graphs
    items = (("tr", t_tr), ("rot", t_rot), ("tor", t_tor), ("sc_tor", t_sc_tor))

    def const(n):
        made, out = {}, {}
        for k, v in items:
            if torch.is_tensor(v):
                out[k] = v * torch.ones(n, device=device)
                continue
            key = float(v)
            if key not in made:
                made[key] = torch.full((n,), key, device=device)
            out[k] = made[key]
        return out

    for nt in ("ligand", "receptor", "atom"):
        if nt in batch:
            batch[nt].node_t = const(batch[nt].num_nodes)
    batch.complex_t = const(b)
    # what the score model would otherwise have to ask the device (a host synchronisation per forward): do all receptor-side
    # nodes sit at ONE translation time?  The hint names the tensors it describes (identity + version counter), so that time
    # tensors edited or replaced afterwards are not covered by it.
    if "receptor" in batch and "atom" in batch:
        ts = (batch["receptor"].node_t["tr"], batch["atom"].node_t["tr"])
        batch.ddp_time_hint = (tuple(id(t) for t in ts), tuple(t._version for t in ts), not torch.is_tensor(t_tr))
    return batch
