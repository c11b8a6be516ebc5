"""Synthetic 3dpf-shaped complex graphs (the benchmark input).

Geometry (ligand heavy atoms + bonds + rotatable-bond masks, pocket residues CA, pocket heavy atoms,
receptor CA graph, atom->residue map, flexible side-chain chi bonds) is REAL 3dpf data, captured once as a
data fixture (`assets/3dpf_geometry.npz`, produced by oracle/make_3dpf_geometry.py from the reference's
example_data/).  Categorical node features and the ESM block are drawn at random inside the reference's
feature vocabularies (datasets/process_mols.py:69-97) because the real featurisers need rdkit/biopython/ESM,
which are out of scope.  The graph schema is the reference's (datasets/process_mols.py:450-453,695-723,895-912).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .batch import HeteroBatch, Store

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

LIG_FEATURE_DIMS = [119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2]   # process_mols.py:69-86
REC_ATOM_FEATURE_DIMS = [38, 119, 23, 38]                                   # process_mols.py:88-93
REC_RESIDUE_FEATURE_DIMS = [38]                                             # process_mols.py:95-97
ESM_DIM = 1280

_ELEMENT_Z = {"C": 6, "N": 7, "O": 8, "S": 16, "P": 15, "F": 9, "CL": 17, "BR": 35, "I": 53}


def load_3dpf_geometry():
    with np.load(os.path.join(ASSETS, "3dpf_geometry.npz")) as z:
        return {k: z[k] for k in z.files}


def make_3dpf_complex(seed: int = 0, flexible_sidechains: bool = True, n_lig=None, n_rec=None, n_atom=None) -> HeteroBatch:
    """One complex graph (CPU tensors).  Optional n_* truncate the complex (small parity cases)."""
    g = load_3dpf_geometry()
    rng = torch.Generator().manual_seed(seed)

    def randcat(n, dims):
        return torch.stack([torch.randint(0, d, (n,), generator=rng) for d in dims], 1)

    data = HeteroBatch()
    # ---- ligand
    lig_pos = torch.from_numpy(g["lig_pos"]).float()
    ei = torch.from_numpy(g["lig_edge_index"]).long()
    ea = torch.from_numpy(g["lig_edge_attr"]).float()
    em = torch.from_numpy(g["lig_edge_mask"]).bool()
    mr = g["lig_mask_rotate"].copy()
    if n_lig is not None and n_lig < lig_pos.shape[0]:
        keep = (ei[0] < n_lig) & (ei[1] < n_lig)
        # rotatable-bond masks of a truncated ligand: drop masks of dropped bonds, clip columns
        rot_keep = keep[em]
        lig_pos, ei, ea, em = lig_pos[:n_lig], ei[:, keep], ea[keep], em[keep]
        mr = mr[rot_keep.numpy()][:, :n_lig]
    x = randcat(lig_pos.shape[0], LIG_FEATURE_DIMS)
    z = torch.tensor([_ELEMENT_Z.get(str(e).upper(), 6) - 1 for e in g["lig_elem"][:lig_pos.shape[0]]])
    x[:, 0] = z
    data["ligand"] = Store(x=x, pos=lig_pos, edge_mask=em, mask_rotate=mr)
    data["ligand", "ligand"] = Store(edge_index=ei, edge_attr=ea)

    # ---- receptor residues
    rec_pos = torch.from_numpy(g["rec_pos"]).float()
    rei = torch.from_numpy(g["rec_edge_index"]).long()
    atom_pos = torch.from_numpy(g["atom_pos"]).float()
    atom_res = torch.from_numpy(g["atom_res"]).long()
    fe = torch.from_numpy(g["flex_edge_idx"]).long()
    fs = torch.from_numpy(g["flex_subcomponents"]).long()
    fm = torch.from_numpy(g["flex_subcomponents_mapping"]).long()
    if n_rec is not None and n_rec < rec_pos.shape[0]:
        # keep the n_rec residues whose CA is closest to the ligand centroid (original order preserved)
        d = (rec_pos - lig_pos.mean(0, keepdim=True)).norm(dim=1)
        keep_r = torch.zeros(rec_pos.shape[0], dtype=torch.bool)
        keep_r[torch.topk(d, n_rec, largest=False).indices] = True
        new_r = torch.cumsum(keep_r.long(), 0) - 1
        rec_pos = rec_pos[keep_r]
        ekeep = keep_r[rei[0]] & keep_r[rei[1]]
        rei = new_r[rei[:, ekeep]]
        keep_a = keep_r[atom_res]
        new_a = torch.cumsum(keep_a.long(), 0) - 1
        atom_pos, atom_res = atom_pos[keep_a], new_r[atom_res[keep_a]]
        if fe.numel():
            ok = keep_a[fe].all(1) & torch.tensor([bool(keep_a[fs[a:b]].all()) for a, b in fm.tolist()])
            new_s, new_m = [], []
            for (a, b), k in zip(fm.tolist(), ok.tolist()):
                if k:
                    new_m.append([len(new_s), len(new_s) + (b - a)])
                    new_s += new_a[fs[a:b]].tolist()
            fe = new_a[fe[ok]]
            fs = torch.tensor(new_s, dtype=torch.long)
            fm = torch.tensor(new_m, dtype=torch.long).reshape(-1, 2)
    if n_atom is not None and n_atom < atom_pos.shape[0]:
        atom_pos, atom_res = atom_pos[:n_atom], atom_res[:n_atom]
    na = atom_pos.shape[0]
    if fe.numel() and (fe.max() >= na or fs.max() >= na):
        ok = (fe < na).all(1) & torch.tensor([bool((fs[a:b] < na).all()) for a, b in fm.tolist()])
        new_s, new_m = [], []
        for (a, b), k in zip(fm.tolist(), ok.tolist()):
            if k:
                new_m.append([len(new_s), len(new_s) + (b - a)])
                new_s += fs[a:b].tolist()
        fe, fs, fm = fe[ok], torch.tensor(new_s, dtype=torch.long), torch.tensor(new_m, dtype=torch.long).reshape(-1, 2)
    nr = rec_pos.shape[0]
    res_x = torch.cat([randcat(nr, REC_RESIDUE_FEATURE_DIMS).float(),
                       torch.randn(nr, ESM_DIM, generator=rng)], 1)
    data["receptor"] = Store(x=res_x, pos=rec_pos)
    data["receptor", "receptor"] = Store(edge_index=rei)
    ax = randcat(na, REC_ATOM_FEATURE_DIMS)
    data["atom"] = Store(x=ax, pos=atom_pos)
    data["atom", "receptor"] = Store(edge_index=torch.stack([torch.arange(na), atom_res], 0))
    if flexible_sidechains and fe.shape[0] > 0:
        st = Store(edge_idx=fe, subcomponents=fs, subcomponentsMapping=fm)
        st.num_nodes = fe.shape[0]
        data["flexResidues"] = st
    data.num_graphs = 1
    data.name = "3dpf"
    return data
