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


def _esm_rows(esm_embeddings, name):
    """The ESM embedding stored for a complex (None if there is none): a tensor / array, or a list of per-chain ones."""
    if esm_embeddings is None:
        return None
    if isinstance(esm_embeddings, dict):
        return esm_embeddings.get(name)
    for ext in (".pt", ".npy"):
        p = os.path.join(esm_embeddings, name + ext)
        if os.path.exists(p):
            return torch.load(p, weights_only=True) if ext == ".pt" else np.load(p)
    return None


def build_row_graph(row: Dict, esm_embeddings=None, root: str = "", allow_zero_esm: bool = False, **graph_kwargs):
    """One csv row -> complex graph (receptor.x = [residue index | 1280 ESM columns]).  The stored embedding may cover the
    whole structure (per chain or concatenated): it is sliced with the kept-residue indices like the reference does
    (inputs.slice_lm_embeddings).  A row without an embedding is an ERROR (-> the row is reported as skipped) unless
    `allow_zero_esm` asks for a zero block, which is out of the model's training distribution and is announced loudly."""
    lig = row["ligand"]
    if not lig.lower().endswith(".sdf"):
        raise NotImplementedError(f"ligand '{lig}': only SDF files are read without rdkit")
    with open(os.path.join(root, row["experimental_protein"])) as f:
        pdb_text = f.read()
    with open(os.path.join(root, lig)) as f:
        sdf_text = f.read()
    e = _esm_rows(esm_embeddings, row["complex_name"])
    if e is None and not allow_zero_esm:
        raise ValueError(f"{row['complex_name']}: no ESM embedding found (pass allow_zero_esm=True to run on a zero block)")
    g = I.build_complex_graph(pdb_text, sdf_text, name=row["complex_name"], pocket_center=row.get("pocket_center"),
                              flexible_sidechains=row.get("flexible_sidechains"), lm_embeddings=e, **graph_kwargs)
    if e is None:
        import warnings
        warnings.warn(f"{row['complex_name']}: no ESM embedding - the receptor gets a ZERO language-model block "
                      f"(out-of-distribution input; allow_zero_esm=True)", RuntimeWarning, stacklevel=2)
        n_res = g["receptor"].x.shape[0]
        g["receptor"].x = torch.cat([g["receptor"].x.float()[:, :1], torch.zeros(n_res, 1280)], 1)
    return g


def run_csv(csv_path: str, model, device, *, confidence_model=None, samples_per_complex: int = 40, inference_steps: int = 20,
            esm_embeddings=None, root: str = "", seed: int = 0, rank: int = 0, world: int = 1, shard: str = "samples",
            dist=None, sampler_cfg: Optional[SamplerConfig] = None, graph_kwargs: Optional[Dict] = None,
            allow_zero_esm: bool = False) -> List[ComplexResult]:
    """See the module docstring.  `dist`: an initialised torch.distributed module (world > 1 and shard == "samples").
    Returns one ComplexResult per csv row (on every rank; with shard == "complexes" only this rank's rows are filled).

    A row that fails on ANY rank (unreadable file, unsupported ligand format, missing ESM embedding, a parsing error: every
    Exception, like the reference's per-complex try / except, inference.py:282-287) is skipped on ALL ranks: with sample
    sharding the ranks agree on the outcome (one all_reduce of an ok flag per row) before anyone enters the sampling loop and
    its final all_gather, and once more after sampling and the confidence pass (a sampling-time failure is rank-local: every rank
    holds different poses), so a rank-local failure cannot leave the others waiting in a collective."""
    dev = torch.device(device)
    if dev.type == "cuda":      # kernels are queued on the CURRENT device's stream: make `device` current for the whole run
        with torch.cuda.device(dev):
            return _run_csv(csv_path, model, dev, confidence_model, samples_per_complex, inference_steps, esm_embeddings, root,
                            seed, rank, world, shard, dist, sampler_cfg, graph_kwargs, allow_zero_esm)
    return _run_csv(csv_path, model, dev, confidence_model, samples_per_complex, inference_steps, esm_embeddings, root, seed,
                    rank, world, shard, dist, sampler_cfg, graph_kwargs, allow_zero_esm)


def _all_ok(dist, ok: bool, device) -> bool:
    """True iff `ok` on every rank."""
    on_gpu = device.type == "cuda" and str(dist.get_backend()).lower() == "nccl"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_gpu else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def _run_csv(csv_path, model, device, confidence_model, samples_per_complex, inference_steps, esm_embeddings, root, seed, rank,
             world, shard, dist, sampler_cfg, graph_kwargs, allow_zero_esm) -> List[ComplexResult]:
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
        split = shard == "samples" and world > 1
        g = None
        try:
            g = build_row_graph(row, esm_embeddings, root, allow_zero_esm=allow_zero_esm, **(graph_kwargs or {}))
        except Exception as e:      # noqa: BLE001 - the reference skips a failing complex and goes on (inference.py:282-287)
            res.skipped = f"{type(e).__name__}: {e}"
        if split and not _all_ok(dist, g is not None, device):
            res.skipped = res.skipped or "skipped: the row failed on another rank"
            continue
        if g is None:
            continue
        flex = bool(getattr(model, "flexible_sidechains", False)) and len(g["flexResidues"]) > 0
        cfg = sampler_cfg or SamplerConfig(inference_steps=inference_steps, flexible_sidechains=flex)
        n = samples_per_complex
        sl = slice(rank * n // world, (rank + 1) * n // world) if split else slice(0, n)
        # Sampling can fail on ONE rank only (each rank holds other poses: a truncated ligand<-atom list - DdpError after the run's
        # final synchronisation -, DDP_ELIMIT, out of memory): like the reference (inference.py:282-287) the complex is then
        # skipped - on EVERY rank, agreed before anyone enters the gathers below
        lig = conf = smp = None
        try:
            smp = Sampler(model, g, n, device, cfg, seed=seed + i, sample_slice=sl)
            smp.randomize()
            smp.run(schedule)
            lig = smp.lig_pos
            if confidence_model is not None:
                conf, _ = smp.confidence(confidence_model)
        except Exception as e:      # noqa: BLE001
            res.skipped = f"{type(e).__name__}: {e}"
            lig = None
        finally:
            if smp is not None and hasattr(smp, "close"):
                if lig is not None:
                    lig = lig.clone()       # (the poses live in the sampler's buffers)
                smp.close()                 # the captured step's memory goes back to the shared pool before the next complex
        if split and not _all_ok(dist, lig is not None, device):
            res.skipped = res.skipped or "skipped: sampling failed on another rank"
            continue
        if lig is None:
            continue
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
