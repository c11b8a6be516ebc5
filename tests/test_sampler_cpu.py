"""CPU tests of the caller side: batched pose update vs golden vectors produced by the reference's own
utils/diffusion_utils.py / utils/torsion.py / utils/geometry.py, perturbation formulas, and shard invariance of the
sample-sharded sampler over a 2-rank gloo group (stub score function - the real model is HIP only)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from diffdock_pocket_amd import sampler as S
from diffdock_pocket_amd.diffusion import get_t_schedule
from diffdock_pocket_amd.synthetic import make_3dpf_complex

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler_pose_update.pt")


def test_pose_update_matches_reference_functions():
    g = torch.load(GOLD, weights_only=False)
    base = make_3dpf_complex(seed=0, flexible_sidechains=True)
    n = g["tr"].shape[0]
    em = base["ligand"].edge_mask
    bonds = base["ligand", "ligand"].edge_index.t()[em]
    mr = torch.as_tensor(base["ligand"].mask_rotate)
    fr = base["flexResidues"]
    atom0 = base["atom"].pos.unsqueeze(0).repeat(n, 1, 1)
    atom = S.apply_sidechain_torsions(atom0, fr.edge_idx, fr.subcomponents, fr.subcomponentsMapping, g["sc"])
    lig = S.modify_conformer(g["lig_start"], g["tr"], g["rot"], g["tor"], bonds, mr)
    assert float((S.rotvec_to_matrix(g["rot"]) - g["rot_mat"]).abs().max()) < 1e-6
    assert float((atom - g["atom_out"]).abs().max()) < 2e-4       # reference goes through float64 numpy/scipy
    assert float((lig - g["lig_out"]).abs().max()) < 2e-4


def test_kabsch_recovers_rigid_motion():
    torch.manual_seed(0)
    A = torch.randn(5, 20, 3)
    R = S.rotvec_to_matrix(torch.randn(5, 3))
    t = torch.randn(5, 1, 3)
    B = A @ R.transpose(1, 2) + t
    Rk, tk = S.kabsch(A, B)
    assert torch.allclose(Rk, R, atol=1e-5) and torch.allclose(tk, t, atol=1e-4)


class StubModel:
    """Deterministic stand-in for the score model (pure function of the batch), CPU only, for sharding tests."""

    def __init__(self, T, S_):
        self.T, self.S = T, S_

    def __call__(self, b):
        B = b.num_graphs
        lp = b["ligand"].pos.reshape(B, -1, 3)
        ap = b["atom"].pos.reshape(B, -1, 3)
        c = lp.mean(1)
        tr = -0.05 * c
        rot = 0.02 * torch.stack([c[:, 1], -c[:, 0], c[:, 2]], 1)
        tor = 0.01 * lp[:, :self.T, 0].reshape(-1)
        sc = 0.01 * ap[:, :self.S, 1].reshape(-1)
        return tr, rot, tor, sc


def _run(n_total, sl, seed=3, steps=4):
    g = make_3dpf_complex(seed=0, flexible_sidechains=True, n_rec=20)
    T, S_ = int(g["ligand"].edge_mask.sum()), int(g["flexResidues"].edge_idx.shape[0])
    cfg = S.SamplerConfig(inference_steps=steps)
    smp = S.Sampler(StubModel(T, S_), g, n_total, torch.device("cpu"), cfg, seed=seed, sample_slice=sl)
    smp.randomize()
    return smp.run(get_t_schedule(steps))


def _worker(rank, world, n_total, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = n_total // world
    lig, atom = _run(n_total, slice(rank * per, (rank + 1) * per))
    out = [torch.empty_like(lig) for _ in range(world)]
    dist.all_gather(out, lig.contiguous())
    if rank == 0:
        q.put(torch.cat(out, 0))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_is_shard_invariant():
    """Samples are sharded over ranks (no data-path collective), noise is drawn for the whole job and sliced; the
    gathered poses equal a single-process run bit for bit."""
    n_total = 6
    full, _ = _run(n_total, slice(0, n_total))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:          # a port that is free right now (a fixed one may still be in TIME_WAIT)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, n_total, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert gathered.shape == full.shape
    assert torch.equal(gathered, full)


def test_sampler_steps_match_the_reference_loop():
    """Sampler.step (scores -> SDE perturbation with the low-temperature branch -> side-chain and ligand pose update) against
    the reference's OWN loop utils/sampling.py:93-251, run unmodified for three steps on three poses with a stub score function
    of the positions and a seeded global RNG (tests/golden/sampler_loop.pt, oracle/make_golden_sampler.py).  The same seed on
    the Sampler's generator gives the same normal draws in the same order (tr, rot, tor, side chains), so the trajectories
    agree to the fp32 rounding of the pose update (the reference rotates in float64 numpy / scipy)."""
    from oracle.make_golden_sampler import LOOP_N, loop_inputs, stub_scores
    gold = torch.load(os.path.join(os.path.dirname(GOLD), "sampler_loop.pt"), weights_only=True)
    base, graphs = loop_inputs()
    assert torch.equal(torch.stack([d["ligand"].pos for d in graphs]), gold["lig_start"])
    T, S_ = int(base["ligand"].edge_mask.sum()), int(base["flexResidues"].edge_idx.shape[0])
    steps = gold["steps"]
    smp = S.Sampler(lambda b: stub_scores(b, T, S_), base, LOOP_N, torch.device("cpu"), S.SamplerConfig(inference_steps=steps),
                    seed=gold["seed"])
    smp.lig_pos, smp.atom_pos = gold["lig_start"].clone(), gold["atom_start"].clone()
    sched = get_t_schedule(steps)
    for i in range(steps):
        assert float((smp.lig_pos - gold["lig_traj"][i]).abs().max()) < 5e-4, i      # poses at the start of step i
        smp.step(i, sched)
    assert float((smp.lig_pos - gold["lig_out"]).abs().max()) < 5e-4
    assert float((smp.atom_pos - gold["atom_out"]).abs().max()) < 5e-4
    assert float((gold["lig_out"] - gold["lig_start"]).abs().max()) > 1.0            # the ligands moved by angstroms
    assert float((gold["atom_out"] - gold["atom_start"]).abs().max()) > 0.1          # and side chains turned
