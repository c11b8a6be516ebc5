"""Batch driver over a `protein_ligand_csv` (BASELINE configs[3]): the loop of reference inference.py:459-493 around the
sampler, for one process per GPU.

    complex_name,experimental_protein,ligand,pocket_center_x,pocket_center_y,pocket_center_z,flexible_sidechains
    (reference data/protein_ligand_example.csv; datasets/pdbbind.py:1005-1066 `load_protein_ligand_df`: rows without a
    ligand / protein are dropped, empty pocket / side-chain cells mean "not given")

Per row: the complex graph from PDB / SDF text (inputs.build_complex_graph - no rdkit / biopython), precomputed ESM rows
(`esm_embeddings`: a {complex_name: [n_residues, 1280]} mapping, or a directory of `<complex_name>.pt` / `.npy` files; the ESM
language model itself is out of scope), `samples_per_complex` poses through Sampler, the confidence
pass and the ranking of reference inference.py:212-219.

Multi-GPU: the SAMPLES of every complex are sharded over the ranks - rank r owns samples
[r*N/R, (r+1)*N/R) of the job's seeded noise stream, no collective inside the denoising loop - and the final ligand poses
and confidences are gathered once per complex (`torch.distributed.all_gather`, RCCL on the GPUs).  The reference shards
COMPLEXES over a process pool instead (inference.py:468, `np.array_split`): `shard="complexes"` does that (no collective at
all; rank r gets rows r::R... contiguous chunks as np.array_split gives them).

Only SDF ligands are read (the reference also accepts SMILES / mol2 through rdkit: out of scope, such rows are reported as
skipped, like the reference's per-complex try / except that returns 0 and goes on, inference.py:282-287)."""
from __future__ import annotations

import csv
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import inputs as I
from .diffusion import get_t_schedule
from .sampler import Sampler, SamplerConfig


@dataclass
class ComplexResult:
    name: str
    ligand_pos: Optional[torch.Tensor] = None      # [N, n_lig, 3], pocket-centred coordinates, ranked best first
    confidence: Optional[torch.Tensor] = None      # [N] (or [N, k]), same order
    order: Optional[torch.Tensor] = None           # sample indices in ranked order
    original_center: Optional[torch.Tensor] = None
    skipped: Optional[str] = None                  # reason, if the row could not be processed


def _none(v):
    return None if v is None or str(v).strip() == "" or str(v).strip().lower() in ("nan", "none") else v


def load_protein_ligand_csv(path: str) -> List[Dict]:
    """Rows of the csv with the cleaning of `load_protein_ligand_df` (datasets/pdbbind.py:1000-1066)."""
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if _none(r.get("ligand")) is None or _none(r.get("experimental_protein")) is None:
                continue
            c = [_none(r.get(f"pocket_center_{a}")) for a in "xyz"]
            rows.append({"complex_name": r["complex_name"], "experimental_protein": r["experimental_protein"], "ligand": r["ligand"],
                         "pocket_center": [float(v) for v in c] if all(v is not None for v in c) else None,
                         "flexible_sidechains": _none(r.get("flexible_sidechains"))})
    return rows


def _esm_rows(esm_embeddings, name, n_res):
    if esm_embeddings is None:
        return None
    if isinstance(esm_embeddings, dict):
        e = esm_embeddings.get(name)
    else:
        e = None
        for ext in (".pt", ".npy"):
            p = os.path.join(esm_embeddings, name + ext)
            if os.path.exists(p):
                e = torch.load(p, weights_only=True) if ext == ".pt" else np.load(p)
                break
    if e is None:
        return None
    e = np.asarray(e.cpu() if torch.is_tensor(e) else e, dtype=np.float32)
    if e.shape[0] != n_res or e.shape[1] != 1280:
        raise ValueError(f"{name}: ESM block {e.shape}, receptor has {n_res} residues")
    return e


def build_row_graph(row: Dict, esm_embeddings=None, root: str = "", **graph_kwargs):
    """One csv row -> complex graph (receptor.x = [residue index | 1280 ESM columns]; zeros if no embedding is given)."""
    lig = row["ligand"]
    if not lig.lower().endswith(".sdf"):
        raise NotImplementedError(f"ligand '{lig}': only SDF files are read without rdkit")
    with open(os.path.join(root, row["experimental_protein"])) as f:
        pdb_text = f.read()
    with open(os.path.join(root, lig)) as f:
        sdf_text = f.read()
    g = I.build_complex_graph(pdb_text, sdf_text, name=row["complex_name"], pocket_center=row.get("pocket_center"),
                              flexible_sidechains=row.get("flexible_sidechains"), **graph_kwargs)
    n_res = g["receptor"].x.shape[0]
    e = _esm_rows(esm_embeddings, row["complex_name"], n_res)
    block = torch.from_numpy(e) if e is not None else torch.zeros(n_res, 1280)
    g["receptor"].x = torch.cat([g["receptor"].x.float()[:, :1], block], 1)
    return g


def run_csv(csv_path: str, model, device, *, confidence_model=None, samples_per_complex: int = 40, inference_steps: int = 20,
            esm_embeddings=None, root: str = "", seed: int = 0, rank: int = 0, world: int = 1, shard: str = "samples",
            dist=None, sampler_cfg: Optional[SamplerConfig] = None, graph_kwargs: Optional[Dict] = None) -> List[ComplexResult]:
    """See the module docstring.  `dist`: an initialised torch.distributed module (world > 1 and shard == "samples").
    Returns one ComplexResult per csv row (on every rank; with shard == "complexes" only this rank's rows are filled)."""
    rows = load_protein_ligand_csv(csv_path)
    if shard not in ("samples", "complexes"):
        raise ValueError(shard)
    mine = range(len(rows))
    if shard == "complexes" and world > 1:
        mine = np.array_split(np.arange(len(rows)), world)[rank].tolist()      # inference.py:468
    out: List[ComplexResult] = []
    schedule = get_t_schedule(inference_steps)
    for i, row in enumerate(rows):
        res = ComplexResult(name=row["complex_name"])
        out.append(res)
        if i not in mine:
            continue
        try:
            g = build_row_graph(row, esm_embeddings, root, **(graph_kwargs or {}))
        except (NotImplementedError, OSError, ValueError) as e:      # the reference skips a failing complex and goes on
            res.skipped = f"{type(e).__name__}: {e}"
            continue
        flex = bool(getattr(model, "flexible_sidechains", False)) and len(g["flexResidues"]) > 0
        cfg = sampler_cfg or SamplerConfig(inference_steps=inference_steps, flexible_sidechains=flex)
        n = samples_per_complex
        split = shard == "samples" and world > 1
        sl = slice(rank * n // world, (rank + 1) * n // world) if split else slice(0, n)
        smp = Sampler(model, g, n, device, cfg, seed=seed + i, sample_slice=sl)
        smp.randomize()
        smp.run(schedule)
        lig = smp.lig_pos
        conf = None
        if confidence_model is not None:
            conf, _ = smp.confidence(confidence_model)
        if split:
            sizes = [(r + 1) * n // world - r * n // world for r in range(world)]
            lig = _gather_rows(dist, lig, sizes)
            if conf is not None:
                conf = _gather_rows(dist, conf, sizes)
        if conf is not None:      # reference inference.py:212-219: descending confidence (first column of a multi-output head)
            key = conf[:, 0] if conf.dim() == 2 else conf
            order = torch.argsort(key, descending=True)
        else:
            order = torch.arange(lig.shape[0], device=lig.device)
        res.order = order.cpu()
        res.ligand_pos = lig[order].cpu()
        res.confidence = conf[order].cpu() if conf is not None else None
        res.original_center = getattr(g, "original_center", None)
    return out


def _gather_rows(dist, t: torch.Tensor, sizes: Sequence[int]) -> torch.Tensor:
    """all_gather of per-rank row blocks of different lengths (padded to the longest)."""
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    buf[: t.shape[0]] = t
    parts = [torch.empty_like(buf) for _ in sizes]
    dist.all_gather(parts, buf.contiguous())
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)
