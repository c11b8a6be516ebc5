"""CPU restatement (plain PyTorch, fp32 or fp64) of the reference's all-atom score-model forward.

TEST INFRASTRUCTURE (oracle/): the checker, never the product.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this.  It runs anywhere torch runs (no e3nn / torch_scatter /
torch_cluster): third-party semantics come from oracle/thirdparty.py (PARITY UNPINNED there, see its
header); everything that is written in the reference's own files is restated here op-for-op and pinned by
the golden vectors in tests/golden/ (generated from the reference's files by oracle/make_golden.py).

Follows (reference file:line):
  models/all_atom_score_model.py:238-436  forward            -> OracleScoreModel.forward
  models/all_atom_score_model.py:444-636  graph builders     -> _lig_graph/_rec_graph/_atom_graph/_cross_graphs/...
  models/all_atom_score_model.py:638-652  get_sc_tor_bonds   -> _sc_tor_bonds
  models/score_model.py:108-125           TensorProductConvLayer.forward -> _conv
  models/layers.py:34-85                  FasterTensorProduct.forward    -> faster_tensor_product
  models/score_model.py:74-82 / :39-52    AtomEncoder / OldAtomEncoder   -> _atom_encoder
  models/score_model.py:661-671           GaussianSmearing               -> gaussian_smearing
  utils/diffusion_utils.py:22-34,73-84    t_to_sigma, sinusoidal_embedding
  utils/so3.py:85-89, utils/torus.py:78-82 score_norm lookups (oracle/score_norm.py)

The model is a pure function of (config, state_dict with the reference's key names, batch).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import thirdparty as tp
from .score_norm import ScoreNormTables

LIG_FEATURE_DIMS = [119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2]   # datasets/process_mols.py:69-86
REC_ATOM_FEATURE_DIMS = [38, 119, 23, 38]                                   # :88-93
REC_RESIDUE_FEATURE_DIMS = [38]                                             # :95-97


@dataclass
class OracleConfig:
    """Static configuration = the reference constructor kwargs (all_atom_score_model.py:22-32) that the README
    configuration exercises, plus the sigma ranges consumed by t_to_sigma."""
    ns: int = 16
    nv: int = 4
    num_conv_layers: int = 2
    sigma_embed_dim: int = 32
    distance_embed_dim: int = 32
    cross_distance_embed_dim: int = 32
    in_lig_edge_features: int = 4
    lig_max_radius: float = 5.0
    rec_max_radius: float = 30.0
    cross_max_distance: float = 80.0
    center_max_distance: float = 30.0
    dynamic_max_cross: bool = True
    scale_by_sigma: bool = True
    batch_norm: bool = True
    no_torsion: bool = False
    fixed_center_conv: bool = True
    atom_max_neighbors: Optional[int] = 8
    flexible_sidechains: bool = True
    use_old_atom_encoder: bool = False
    confidence_mode: bool = False
    lm_embedding_type: Optional[str] = "esm"
    embedding_scale: float = 1000.0
    tr_sigma_min: float = 0.1
    tr_sigma_max: float = 5.0
    rot_sigma_min: float = 0.03
    rot_sigma_max: float = 1.55
    tor_sigma_min: float = 0.03
    tor_sigma_max: float = 3.14
    sidechain_tor_sigma_min: float = 0.03
    sidechain_tor_sigma_max: float = 3.14
    smooth_edges: bool = False      # (all_atom_score_model.py:438-442) cosine edge weights on the fc outputs

    def irreps(self, i):
        ns, nv = self.ns, self.nv
        seq = [f"{ns}x0e", f"{ns}x0e+{nv}x1o", f"{ns}x0e+{nv}x1o+{nv}x1e", f"{ns}x0e+{nv}x1o+{nv}x1e+{ns}x0o"]
        return seq[min(i, 3)]


# ------------------------------------------------------------------------------------------------ primitives

def sinusoidal_embedding(t, dim, scale, max_positions=10000):
    half = dim // 2
    w = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(max_positions) / (half - 1)))
    arg = scale * t.float()[:, None] * w[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    return F.pad(emb, (0, 1)) if dim % 2 == 1 else emb


def gaussian_smearing(dist, offset):
    """exp(coeff * (d - mu_k)^2), coeff = -0.5 / (mu_1 - mu_0)^2 with the python-float coeff of the reference."""
    coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
    d = dist.reshape(-1, 1) - offset.reshape(1, -1)
    return torch.exp(coeff * torch.pow(d, 2))


def faster_tensor_product(in_irreps, out_irreps, x, sh, weight):
    """l<=1 Clebsch-Gordan tensor product with per-edge weights (models/layers.py:34-85)."""
    in_irreps, out_irreps = tp.Irreps(in_irreps), tp.Irreps(out_irreps)
    mi = {"0e": 0, "1o": 0, "1e": 0, "0o": 0}
    mo = dict(mi)
    parts = {}
    for m, sl in zip(in_irreps, in_irreps.slices()):
        mi[str(m.ir)] = m.mul
        v = x[:, sl]
        parts[str(m.ir)] = v.reshape(v.shape[0], -1, 3) if m.ir.l == 1 else v
    for m in out_irreps:
        mo[str(m.ir)] = m.mul
    s0, s1 = sh[:, 0], sh[:, 1:4]
    feats = {"0e": [], "1o": [], "1e": [], "0o": []}
    r3, r2 = math.sqrt(3.0), math.sqrt(2.0)
    if "0e" in parts:
        a = parts["0e"]
        feats["0e"].append(a * s0[:, None])
        feats["1o"].append(a[:, :, None] * s1[:, None, :])
    if "1o" in parts:
        a = parts["1o"]
        feats["0e"].append((a * s1[:, None, :]).sum(-1) / r3)
        feats["1o"].append(a * s0[:, None, None])
        feats["1e"].append(torch.linalg.cross(a, s1[:, None, :].expand_as(a), dim=-1) / r2)
    if "1e" in parts:
        a = parts["1e"]
        feats["1o"].append(torch.linalg.cross(a, s1[:, None, :].expand_as(a), dim=-1) / r2)
        feats["1e"].append(a * s0[:, None, None])
        feats["0o"].append((a * s1[:, None, :]).sum(-1) / r3)
    if "0o" in parts:
        a = parts["0o"]
        feats["1e"].append(a[:, :, None] * s1[:, None, :])
        feats["0o"].append(a * s0[:, None])
    shapes = {"0e": (mi["0e"] + mi["1o"], mo["0e"]), "1o": (mi["0e"] + mi["1o"] + mi["1e"], mo["1o"]),
              "1e": (mi["1o"] + mi["1e"] + mi["0o"], mo["1e"]), "0o": (mi["1e"] + mi["0o"], mo["0o"])}
    outs, off = {}, 0
    for key, (u, n) in shapes.items():
        w = weight[:, off:off + u * n].reshape(-1, u, n) / np.sqrt(u) if u * n > 0 else None
        off += u * n
        if not feats[key] or w is None:
            continue
        f = torch.cat(feats[key], dim=1)
        if key in ("0e", "0o"):
            outs[key] = torch.matmul(f.unsqueeze(-2), w).squeeze(-2)
        else:
            outs[key] = (f.unsqueeze(-2) * w.unsqueeze(-1)).sum(-3).reshape(f.shape[0], -1)
    assert off == weight.shape[1]
    return torch.cat([outs[str(m.ir)] for m in out_irreps if m.mul > 0], dim=-1)


def weight_numel_faster(in_irreps, out_irreps):
    in_irreps, out_irreps = tp.Irreps(in_irreps), tp.Irreps(out_irreps)
    mi = {"0e": 0, "1o": 0, "1e": 0, "0o": 0}
    mo = dict(mi)
    for m in in_irreps:
        mi[str(m.ir)] = m.mul
    for m in out_irreps:
        mo[str(m.ir)] = m.mul
    return ((mi["0e"] + mi["1o"]) * mo["0e"] + (mi["0e"] + mi["1o"] + mi["1e"]) * mo["1o"]
            + (mi["1o"] + mi["1e"] + mi["0o"]) * mo["1e"] + (mi["1e"] + mi["0o"]) * mo["0o"])


# ------------------------------------------------------------------------------------------------ the model

class OracleScoreModel:
    def __init__(self, cfg: OracleConfig, state_dict: Dict[str, torch.Tensor], tables: Optional[ScoreNormTables] = None,
                 dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        self.sd = {k: (v.detach().to("cpu").to(dtype) if v.is_floating_point() else v.detach().cpu())
                   for k, v in state_dict.items()}
        self.tables = tables or ScoreNormTables.load()
        self.sh_irreps = "1x0e+1x1o"
        self.tor_sh_irreps = "1x1o+1x2e+1x2o+1x3o"
        self.record: Dict[str, torch.Tensor] = {}
        self._w: Dict[str, object] = {}

    # ---- small layers
    def _lin(self, prefix, x):
        y = x @ self.sd[prefix + ".weight"].T
        b = self.sd.get(prefix + ".bias")
        return y if b is None else y + b

    def _mlp(self, prefix, x):  # Linear(0) -> ReLU -> Dropout -> Linear(3)
        return self._lin(prefix + ".3", torch.relu(self._lin(prefix + ".0", x)))

    def _atom_encoder(self, prefix, x, n_cat):
        emb = 0
        for i in range(n_cat):
            emb = emb + self.sd[f"{prefix}.atom_embedding_list.{i}.weight"][x[:, i].long()]
        if self.cfg.use_old_atom_encoder:  # models/score_model.py:39-52
            n_scalar = self.cfg.sigma_embed_dim
            emb = emb + self._lin(prefix + ".linear", x[:, n_cat:n_cat + n_scalar])
            if (prefix + ".lm_embedding_layer.weight") in self.sd:
                emb = self._lin(prefix + ".lm_embedding_layer", torch.cat([emb, x[:, -1280:]], dim=1))
            return emb
        return self._lin(prefix + ".additional_features_embedder", torch.cat([emb, x[:, n_cat:]], dim=1))

    def _emb(self, t):
        return sinusoidal_embedding(t, self.cfg.sigma_embed_dim, self.cfg.embedding_scale).to(self.dtype)

    def _sh(self, vec):
        return tp.spherical_harmonics(self.sh_irreps, vec, normalize=True, normalization="component")

    def _edge_weight(self, vec, max_norm):
        """get_edge_weight (all_atom_score_model.py:438-442): 1, or with smooth_edges 0.5 (cos(min(|v| pi / max_norm, pi)) + 1)
        per edge; max_norm a number or one value per edge (the dynamic cross cutoff of the edge's graph, :558-560)."""
        if not self.cfg.smooth_edges:
            return 1.0
        nn_ = torch.clip(vec.norm(dim=-1) * math.pi / max_norm, max=math.pi)
        return 0.5 * (torch.cos(nn_) + 1.0).unsqueeze(-1)

    def _conv(self, prefix, in_irreps, out_irreps, node_attr, edge_index, edge_attr, edge_sh, out_nodes=None,
              faster=True, sh_irreps=None, edge_weight=1.0):
        """TensorProductConvLayer.forward (models/score_model.py:108-125), residual=False, reduce='mean'; edge_weight scales
        the fc output of an edge (:114)."""
        if edge_index.numel() == 0:
            return torch.tensor(0, dtype=node_attr.dtype)
        recv, src = edge_index[0], edge_index[1]
        w = self._mlp(prefix + ".fc", edge_attr) * edge_weight
        if faster:
            msg = faster_tensor_product(in_irreps, out_irreps, node_attr[src], edge_sh, w)
        else:
            fctp = tp.FullyConnectedTensorProduct(in_irreps, sh_irreps, out_irreps, shared_weights=False)
            assert fctp.weight_numel == w.shape[1]
            msg = fctp(node_attr[src], edge_sh, w)
        n_out = int(out_nodes) if out_nodes is not None else node_attr.shape[0]
        out = tp.scatter(msg, recv, dim=0, dim_size=n_out, reduce="mean")
        if self.cfg.batch_norm:
            out = tp.batch_norm_eval(out_irreps, out, self.sd[prefix + ".batch_norm.running_mean"],
                                     self.sd[prefix + ".batch_norm.running_var"],
                                     self.sd[prefix + ".batch_norm.weight"], self.sd[prefix + ".batch_norm.bias"])
        return out

    # ---- graph builders (all_atom_score_model.py:444-636)
    def _lig_graph(self, data):
        c, lig = self.cfg, data["ligand"]
        pos = lig.pos.to(self.dtype)
        lig.node_sigma_emb = self._emb(lig.node_t["tr"])
        radius_edges = tp.radius_graph(lig.pos, c.lig_max_radius, lig.batch)
        ei = torch.cat([data["ligand", "ligand"].edge_index, radius_edges], 1).long()
        ea = torch.cat([data["ligand", "ligand"].edge_attr.to(self.dtype),
                        torch.zeros(radius_edges.shape[1], c.in_lig_edge_features, dtype=self.dtype)], 0)
        ea = torch.cat([ea, lig.node_sigma_emb[ei[0]]], 1)
        node_attr = torch.cat([lig.x.to(self.dtype), lig.node_sigma_emb], 1)
        vec = pos[ei[1]] - pos[ei[0]]
        ea = torch.cat([ea, gaussian_smearing(vec.norm(dim=-1), self.sd["lig_distance_expansion.offset"])], 1)
        self._w["ll"] = self._edge_weight(vec, c.lig_max_radius)      # (:482)
        return node_attr, ei, ea, self._sh(vec)

    def _rec_graph(self, data):
        rec = data["receptor"]
        pos = rec.pos.to(self.dtype)
        rec.node_sigma_emb = self._emb(rec.node_t["tr"])
        node_attr = torch.cat([rec.x.to(self.dtype), rec.node_sigma_emb], 1)
        ei = data["receptor", "receptor"].edge_index.long()
        vec = pos[ei[1]] - pos[ei[0]]
        ea = torch.cat([rec.node_sigma_emb[ei[0]],
                        gaussian_smearing(vec.norm(dim=-1), self.sd["rec_distance_expansion.offset"])], 1)
        self._w["rr"] = self._edge_weight(vec, self.cfg.rec_max_radius)   # (:509)
        return node_attr, ei, ea, self._sh(vec)

    def _atom_graph(self, data):
        c, atom = self.cfg, data["atom"]
        pos = atom.pos.to(self.dtype)
        atom.node_sigma_emb = self._emb(atom.node_t["tr"])
        node_attr = torch.cat([atom.x.to(self.dtype), atom.node_sigma_emb], 1)
        ei = tp.knn_graph(atom.pos, k=c.atom_max_neighbors if c.atom_max_neighbors else 32, batch=atom.batch)
        data["atom", "atom"].edge_index = ei
        vec = pos[ei[1]] - pos[ei[0]]
        ea = torch.cat([atom.node_sigma_emb[ei[0]],
                        gaussian_smearing(vec.norm(dim=-1), self.sd["lig_distance_expansion.offset"])], 1)
        self._w["aa"] = self._edge_weight(vec, c.lig_max_radius)      # (:535)
        return node_attr, ei, ea, self._sh(vec)

    def _cross_graphs(self, data, cutoff):
        c = self.cfg
        lig, rec, atom = data["ligand"], data["receptor"], data["atom"]
        lp, rp, ap = lig.pos.to(self.dtype), rec.pos.to(self.dtype), atom.pos.to(self.dtype)
        if torch.is_tensor(cutoff):
            lr = tp.radius(rec.pos / cutoff[rec.batch], lig.pos / cutoff[lig.batch], 1, rec.batch, lig.batch,
                           max_num_neighbors=10000)
        else:
            lr = tp.radius(rec.pos, lig.pos, cutoff, rec.batch, lig.batch, max_num_neighbors=10000)
        off_x = self.sd["cross_distance_expansion.offset"]
        v = rp[lr[1]] - lp[lr[0]]
        lr_attr = torch.cat([lig.node_sigma_emb[lr[0]], gaussian_smearing(v.norm(dim=-1), off_x)], 1)
        lr_sh = self._sh(v)
        # (:558-560) the cutoff of the ligand atom's graph, or the fixed one
        self._w["lr"] = self._edge_weight(v, cutoff[lig.batch[lr[0]]].squeeze().to(self.dtype) if torch.is_tensor(cutoff) else cutoff)
        la = tp.radius(atom.pos, lig.pos, c.lig_max_radius, atom.batch, lig.batch, max_num_neighbors=10000)
        v = ap[la[1]] - lp[la[0]]
        la_attr = torch.cat([lig.node_sigma_emb[la[0]], gaussian_smearing(v.norm(dim=-1), off_x)], 1)
        la_sh = self._sh(v)
        self._w["la"] = self._edge_weight(v, c.lig_max_radius)        # (:571); atom-receptor edges: weight 1 (:580)
        ar = data["atom", "receptor"].edge_index.long()
        v = rp[ar[1]] - ap[ar[0]]
        ar_attr = torch.cat([atom.node_sigma_emb[ar[0]],
                             gaussian_smearing(v.norm(dim=-1), self.sd["rec_distance_expansion.offset"])], 1)
        ar_sh = self._sh(v)
        return lr, lr_attr, lr_sh, la, la_attr, la_sh, ar, ar_attr, ar_sh

    def _center_graph(self, data):
        lig = data["ligand"]
        pos = lig.pos.to(self.dtype)
        b = data.num_graphs
        ei = torch.stack([lig.batch, torch.arange(len(lig.batch))], 0)
        center = torch.zeros((b, 3), dtype=self.dtype).index_add_(0, lig.batch, pos)
        center = center / torch.bincount(lig.batch, minlength=b).unsqueeze(1)
        vec = pos[ei[1]] - center[ei[0]]
        ea = gaussian_smearing(vec.norm(dim=-1), self.sd["center_distance_expansion.offset"])
        ea = torch.cat([ea, lig.node_sigma_emb[ei[1]]], 1)
        return ei, ea, self._sh(vec)

    def _bond_graph(self, pos, pos32, batch, bonds, bond_batch, edge_mlp):
        """build_bond_conv_graph / build_sidechain_conv_graph (:601-636): radius with the default cap of 32."""
        bond_pos = (pos[bonds[0]] + pos[bonds[1]]) / 2
        bond_pos32 = (pos32[bonds[0]] + pos32[bonds[1]]) / 2
        ei = tp.radius(pos32, bond_pos32, self.cfg.lig_max_radius, batch_x=batch, batch_y=bond_batch)
        vec = pos[ei[1]] - bond_pos[ei[0]]
        ea = self._mlp(edge_mlp, gaussian_smearing(vec.norm(dim=-1), self.sd["lig_distance_expansion.offset"]))
        self._w["bond"] = self._edge_weight(vec, self.cfg.lig_max_radius)   # (:614,634)
        return ei, ea, self._sh(vec)

    @staticmethod
    def _sc_tor_bonds(data):
        _, counts = data["atom"].batch.unique(sorted=True, return_counts=True)
        off = counts.cumsum(0)
        off = torch.cat((torch.zeros(1, dtype=off.dtype), off))[:-1].long()
        return off[data["flexResidues"].batch] + data["flexResidues"].edge_idx.T.long()

    def _torsion_head(self, conv_prefix, final_prefix, node_attr, pos, pos32, batch, bonds, bond_batch, edge_mlp, in_irreps):
        c = self.cfg
        ns = c.ns
        ei, ea, esh = self._bond_graph(pos, pos32, batch, bonds, bond_batch, edge_mlp)
        bond_vec = pos[bonds[1]] - pos[bonds[0]]
        bond_attr = node_attr[bonds[0]] + node_attr[bonds[1]]
        bonds_sh = tp.spherical_harmonics("2e", bond_vec, normalize=True, normalization="component")
        ftp = tp.FullTensorProduct(self.sh_irreps, "2e")
        tor_sh = ftp(esh, bonds_sh[ei[0]])
        ea = torch.cat([ea, node_attr[ei[1], :ns], bond_attr[ei[0], :ns]], -1)
        h = self._conv(conv_prefix, in_irreps, f"{ns}x0o+{ns}x0e", node_attr, ei, ea, tor_sh,
                       out_nodes=bonds.shape[1], faster=False, sh_irreps=self.tor_sh_irreps, edge_weight=self._w["bond"])
        if h.dim() == 0:  # no edge at all: the reference would crash in tor_final_layer; keep the literal behaviour
            raise RuntimeError("torsion head has no edges")
        h = torch.tanh(h @ self.sd[final_prefix + ".0.weight"].T) @ self.sd[final_prefix + ".3.weight"].T
        return h.squeeze(1)

    # ---- forward (all_atom_score_model.py:238-436)
    def _forward_confidence(self, data, sig):
        return self.forward(data, _conf=True)

    def forward(self, data, _conf=False):
        c, ns = self.cfg, self.cfg.ns
        conf = _conf
        rec = self.record = {}
        self._w = {}      # edge weights of the edge sets built below (1.0 without smooth_edges)
        sig = {k: data.complex_t[k].to(self.dtype) for k in ("tr", "rot", "tor", "sc_tor")}
        if c.confidence_mode and not conf:   # (:245) sigmas = times; only the cross cutoff uses them
            return self._forward_confidence(data, sig)
        if conf:
            sig = {k: sig[k] for k in sig}
        tr_sigma = c.tr_sigma_min ** (1 - sig["tr"]) * c.tr_sigma_max ** sig["tr"]
        rot_sigma = c.rot_sigma_min ** (1 - sig["rot"]) * c.rot_sigma_max ** sig["rot"]
        tor_sigma = c.tor_sigma_min ** (1 - sig["tor"]) * c.tor_sigma_max ** sig["tor"]
        sc_sigma = c.sidechain_tor_sigma_min ** (1 - sig["sc_tor"]) * c.sidechain_tor_sigma_max ** sig["sc_tor"]

        lig_x, ll, ll_attr, ll_sh = self._lig_graph(data)
        lig_x = self._atom_encoder("lig_node_embedding", lig_x, len(LIG_FEATURE_DIMS))
        ll_attr = self._mlp("lig_edge_embedding", ll_attr)
        rec_x, rr, rr_attr, rr_sh = self._rec_graph(data)
        rec_x = self._atom_encoder("rec_node_embedding", rec_x, len(REC_RESIDUE_FEATURE_DIMS))
        rr_attr = self._mlp("rec_edge_embedding", rr_attr)
        atom_x, aa, aa_attr, aa_sh = self._atom_graph(data)
        atom_x = self._atom_encoder("atom_node_embedding", atom_x, len(REC_ATOM_FEATURE_DIMS))
        aa_attr = self._mlp("atom_edge_embedding", aa_attr)

        # the neighbour search always runs on float32 inputs (as in the reference), also in the fp64 oracle
        t32 = data.complex_t["tr"].float()
        tr_sigma32 = t32 if conf else c.tr_sigma_min ** (1 - t32) * c.tr_sigma_max ** t32
        cutoff = (tr_sigma32 * 3 + 20).unsqueeze(1) if c.dynamic_max_cross else c.cross_max_distance
        lr, lr_attr, lr_sh, la, la_attr, la_sh, ar, ar_attr, ar_sh = self._cross_graphs(data, cutoff)
        lr_attr = self._mlp("lr_edge_embedding", lr_attr)
        la_attr = self._mlp("la_edge_embedding", la_attr)
        ar_attr = self._mlp("ar_edge_embedding", ar_attr)
        rec.update(lig_x0=lig_x, rec_x0=rec_x, atom_x0=atom_x, ll=ll, rr=rr, aa=aa, lr=lr, la=la, ar=ar,
                   ll_attr=ll_attr, ll_sh=ll_sh, aa_attr=aa_attr, aa_sh=aa_sh, lr_attr=lr_attr, lr_sh=lr_sh,
                   la_attr=la_attr, la_sh=la_sh, ar_attr=ar_attr, ar_sh=ar_sh, rr_attr=rr_attr, rr_sh=rr_sh)

        L = c.num_conv_layers
        for l in range(L):
            ii, oi = c.irreps(l), c.irreps(l + 1)
            W = self._w
            cv = lambda k, *a, **kw: self._conv(f"conv_layers.{9 * l + k}", ii, oi, *a, **kw)
            s = lambda x: x[:, :ns]
            nl, na_, nr = lig_x.shape[0], atom_x.shape[0], rec_x.shape[0]
            u0 = cv(0, lig_x, ll, torch.cat([ll_attr, s(lig_x)[ll[0]], s(lig_x)[ll[1]]], -1), ll_sh, edge_weight=W["ll"])
            u1 = cv(1, rec_x, lr, torch.cat([lr_attr, s(lig_x)[lr[0]], s(rec_x)[lr[1]]], -1), lr_sh, out_nodes=nl, edge_weight=W["lr"])
            u2 = cv(2, atom_x, la, torch.cat([la_attr, s(lig_x)[la[0]], s(atom_x)[la[1]]], -1), la_sh, out_nodes=nl, edge_weight=W["la"])
            do_atom = c.flexible_sidechains or l != L - 1
            do_rec = do_atom and l != L - 1
            if do_atom:
                u3 = cv(3, atom_x, aa, torch.cat([aa_attr, s(atom_x)[aa[0]], s(atom_x)[aa[1]]], -1), aa_sh, edge_weight=W["aa"])
                u4 = cv(4, lig_x, torch.flip(la, dims=[0]),
                        torch.cat([la_attr, s(atom_x)[la[1]], s(lig_x)[la[0]]], -1), la_sh, out_nodes=na_, edge_weight=W["la"])
                u5 = cv(5, rec_x, ar, torch.cat([ar_attr, s(atom_x)[ar[0]], s(rec_x)[ar[1]]], -1), ar_sh, out_nodes=na_)
            if do_rec:
                u6 = cv(6, rec_x, rr, torch.cat([rr_attr, s(rec_x)[rr[0]], s(rec_x)[rr[1]]], -1), rr_sh, edge_weight=W["rr"])
                u7 = cv(7, lig_x, torch.flip(lr, dims=[0]),
                        torch.cat([lr_attr, s(rec_x)[lr[1]], s(lig_x)[lr[0]]], -1), lr_sh, out_nodes=nr, edge_weight=W["lr"])
                u8 = cv(8, atom_x, torch.flip(ar, dims=[0]),
                        torch.cat([ar_attr, s(rec_x)[ar[1]], s(atom_x)[ar[0]]], -1), ar_sh, out_nodes=nr)
            d_out = tp.Irreps(oi).dim
            lig_x = F.pad(lig_x, (0, d_out - lig_x.shape[-1])) + u0 + u2 + u1
            if do_atom:
                atom_x = F.pad(atom_x, (0, d_out - atom_x.shape[-1])) + u3 + u4 + u5
            if do_rec:
                rec_x = F.pad(rec_x, (0, d_out - rec_x.shape[-1])) + u6 + u8 + u7
            rec[f"lig_x{l + 1}"], rec[f"atom_x{l + 1}"], rec[f"rec_x{l + 1}"] = lig_x, atom_x, rec_x

        final_irreps = c.irreps(L)
        has_flex = c.flexible_sidechains and ("flexResidues" in data) and len(data["flexResidues"]) > 0
        n_flex = data["flexResidues"].edge_idx.shape[0] if has_flex else 0
        if conf:   # confidence head (:329-353)
            def scal(x):
                return torch.cat([x[:, :ns], x[:, -ns:]], dim=1) if L >= 3 else x[:, :ns]
            cin = tp.scatter_mean(scal(lig_x), data["ligand"].batch, dim=0)
            if c.flexible_sidechains:
                if n_flex > 0:
                    fa = self._sc_tor_bonds(data).unique()
                    sa = tp.scatter_mean(scal(atom_x)[fa], data["atom"].batch[fa], dim=0, dim_size=cin.shape[0])
                else:
                    sa = torch.zeros_like(cin)
                cin = torch.cat([cin, sa], dim=1)
            p = "confidence_predictor"
            def bn(i, x):
                if f"{p}.{i}.running_mean" not in self.sd:
                    return x
                return ((x - self.sd[f"{p}.{i}.running_mean"]) / torch.sqrt(self.sd[f"{p}.{i}.running_var"] + 1e-5)
                        * self.sd[f"{p}.{i}.weight"] + self.sd[f"{p}.{i}.bias"])
            h = torch.relu(bn(1, self._lin(f"{p}.0", cin)))
            h = torch.relu(bn(5, self._lin(f"{p}.4", h)))
            return self._lin(f"{p}.8", h).squeeze(dim=-1)

        # translation / rotation head (:357-384)
        ce, ce_attr, ce_sh = self._center_graph(data)
        ce_attr = self._mlp("center_edge_embedding", ce_attr)
        ce_attr = torch.cat([ce_attr, lig_x[ce[1] if c.fixed_center_conv else ce[0], :ns]], -1)
        g = self._conv("final_conv", final_irreps, "2x1o+2x1e", lig_x, ce, ce_attr, ce_sh, out_nodes=data.num_graphs)
        rec["global_pred"] = g
        tr = g[:, :3] + g[:, 6:9]
        rot = g[:, 3:6] + g[:, 9:]
        data.graph_sigma_emb = self._emb(data.complex_t["tr"])

        def final_mlp(prefix, x):  # Linear(0) -> Dropout -> ReLU -> Linear(3)
            return self._lin(prefix + ".3", torch.relu(self._lin(prefix + ".0", x)))

        tr_norm = torch.linalg.vector_norm(tr, dim=1).unsqueeze(1)
        tr = tr / tr_norm * final_mlp("tr_final_layer", torch.cat([tr_norm, data.graph_sigma_emb], dim=1))
        rot_norm = torch.linalg.vector_norm(rot, dim=1).unsqueeze(1)
        rot = rot / rot_norm * final_mlp("rot_final_layer", torch.cat([rot_norm, data.graph_sigma_emb], dim=1))
        if c.scale_by_sigma:
            tr = tr / tr_sigma.unsqueeze(1)
            rot = rot * torch.from_numpy(self.tables.so3_score_norm(rot_sigma.float().numpy())).float().to(self.dtype).unsqueeze(1)

        # ligand torsion head (:386-408)
        lig = data["ligand"]
        if c.no_torsion or int(lig.edge_mask.sum()) == 0:
            tor = torch.empty(0, dtype=self.dtype)
        else:
            bonds = data["ligand", "ligand"].edge_index[:, lig.edge_mask].long()
            tor = self._torsion_head("tor_bond_conv", "tor_final_layer", lig_x, lig.pos.to(self.dtype), lig.pos, lig.batch,
                                     bonds, lig.batch[bonds[0]], "final_edge_embedding", final_irreps)
            if c.scale_by_sigma:
                es = tor_sigma[lig.batch][data["ligand", "ligand"].edge_index[0]][lig.edge_mask]
                tor = tor * torch.sqrt(torch.tensor(self.tables.torus_score_norm(es.float().numpy())).float()).to(self.dtype)

        # side-chain torsion head (:410-434)
        if n_flex == 0:
            sc = torch.empty(0, dtype=self.dtype)
        else:
            atom = data["atom"]
            bonds = self._sc_tor_bonds(data)
            sc = self._torsion_head("sc_tor_bond_conv", "sc_tor_final_layer", atom_x, atom.pos.to(self.dtype), atom.pos,
                                    atom.batch, bonds, data["flexResidues"].batch, "sidechain_final_edge_embedding",
                                    final_irreps)
            if c.scale_by_sigma:
                es = sc_sigma[data["flexResidues"].batch]
                sc = sc * torch.sqrt(torch.tensor(self.tables.torus_score_norm(es.float().numpy())).float()).to(self.dtype)
        return tr, rot, tor, sc

    __call__ = forward
