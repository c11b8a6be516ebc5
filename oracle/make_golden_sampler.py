"""Golden vectors for the pose update (the caller side of the hot path), produced by the
reference's OWN functions imported under oracle/shim.py (TEST INFRASTRUCTURE; build container only):
  utils/diffusion_utils.py:37-60  modify_conformer        (rigid move + torsions + Kabsch re-alignment)
  utils/diffusion_utils.py:63-70  modify_sidechains       (sequential chi rotations)
  utils/geometry.py:72-86         axis_angle_to_matrix
and, second file, by the reference's OWN reverse-diffusion loop
  utils/sampling.py:70-251        sampling()               (score -> perturbation incl. the low-temperature branch :177-195 ->
                                                            modify_sidechains / modify_conformer per sample)
run unmodified for three steps on three poses of the 3dpf complex with a deterministic stub score function (a pure function of
the batch's positions, so that every step's scores depend on the previous update) and the global RNG seeded; the PyG DataLoader
it imports is replaced by a three-line stand-in that collates with diffdock_pocket_amd.batch.collate.
Usage: python -m oracle.make_golden_sampler
"""
import copy
import os

import numpy as np
import torch

from diffdock_pocket_amd.synthetic import make_3dpf_complex

from . import shim

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "sampler_pose_update.pt")


def main():
    ref = shim.import_reference()
    g = torch.Generator().manual_seed(11)
    n = 4
    base = make_3dpf_complex(seed=0, flexible_sidechains=True)
    T = int(base["ligand"].edge_mask.sum())
    S = int(base["flexResidues"].edge_idx.shape[0])
    tr = torch.randn(n, 3, generator=g) * 0.7
    rot = torch.randn(n, 3, generator=g) * 0.5
    rot[0] = rot[0] * 1e-8                                   # small-angle branch
    tor = (torch.rand(n, T, generator=g) * 2 - 1) * 2.5
    sc = (torch.rand(n, S, generator=g) * 2 - 1) * 2.5
    lig_out, atom_out, rmat = [], [], []
    for i in range(n):
        d = copy.deepcopy(base)
        d["ligand"].pos = d["ligand"].pos + torch.randn(1, 3, generator=g)
        start_lig = d["ligand"].pos.clone()
        ref.diffusion_utils.modify_sidechains(d, sc[i].numpy())
        d = ref.diffusion_utils.modify_conformer(d, tr[i:i + 1], rot[i], tor[i].numpy())
        lig_out.append(d["ligand"].pos.clone())
        atom_out.append(d["atom"].pos.clone())
        rmat.append(ref.geometry.axis_angle_to_matrix(rot[i]))
        if i == 0:
            starts = [start_lig]
        else:
            starts.append(start_lig)
    torch.save({"tr": tr, "rot": rot, "tor": tor, "sc": sc, "lig_start": torch.stack(starts),
                "lig_out": torch.stack(lig_out), "atom_out": torch.stack(atom_out), "rot_mat": torch.stack(rmat)}, OUT)
    print("wrote", OUT)


OUT_LOOP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "sampler_loop.pt")
LOOP_SEED, LOOP_N, LOOP_STEPS = 5, 3, 3


def stub_scores(b, T, S):
    """Deterministic stand-in for the score model, shared with tests/test_sampler_cpu.py (pure function of the positions)."""
    B = b.num_graphs
    lp = b["ligand"].pos.reshape(B, -1, 3)
    ap = b["atom"].pos.reshape(B, -1, 3)
    c = lp.mean(1)
    tr = -0.05 * c + 0.01 * lp[:, 0]
    rot = 0.02 * torch.stack([c[:, 1], -c[:, 0], c[:, 2]], 1)
    tor = 0.01 * lp[:, :T, 0].reshape(-1) - 0.02
    sc = 0.01 * ap[:, :S, 1].reshape(-1) + 0.015
    return tr, rot, tor, sc


def loop_inputs():
    base = make_3dpf_complex(seed=0, flexible_sidechains=True)
    g = torch.Generator().manual_seed(21)
    graphs = []
    for _ in range(LOOP_N):
        d = copy.deepcopy(base)
        d["ligand"].pos = d["ligand"].pos + torch.randn(1, 3, generator=g) * 1.5
        graphs.append(d)
    return base, graphs


def main_loop():
    import argparse
    import functools
    import importlib
    import sys
    from diffdock_pocket_amd.batch import collate
    ref = shim.import_reference()

    class Loader:   # stand-in for torch_geometric.loader.DataLoader (utils/sampling.py:7,100): batches of collated graphs
        def __init__(self, data_list, batch_size=32):
            self.data_list, self.bs = data_list, batch_size

        def __iter__(self):
            for i in range(0, len(self.data_list), self.bs):
                yield collate(self.data_list[i:i + self.bs])

    sys.modules["torch_geometric.loader"].DataLoader = Loader
    sampling_mod = importlib.import_module("utils.sampling")
    sampling_mod.DataLoader = Loader
    base, graphs = loop_inputs()
    T, S = int(base["ligand"].edge_mask.sum()), int(base["flexResidues"].edge_idx.shape[0])
    from diffdock_pocket_amd.sampler import TEMP_PSI, TEMP_SAMPLING, TEMP_SIGMA_DATA
    margs = argparse.Namespace(tr_sigma_min=0.1, tr_sigma_max=5.0, rot_sigma_min=0.03, rot_sigma_max=1.55, tor_sigma_min=0.03,
                               tor_sigma_max=3.14, sidechain_tor_sigma_min=0.03, sidechain_tor_sigma_max=3.14, no_torsion=False,
                               flexible_sidechains=True, all_atoms=True)
    t_to_sigma = functools.partial(ref.diffusion_utils.t_to_sigma, args=margs)
    sched = np.linspace(1, 0, LOOP_STEPS + 1)[:-1]   # = get_t_schedule (utils/diffusion_utils.py:112-117)
    torch.manual_seed(LOOP_SEED)
    starts_l = torch.stack([d["ligand"].pos.clone() for d in graphs])
    starts_a = torch.stack([d["atom"].pos.clone() for d in graphs])
    out, conf, traj, sc_traj = sampling_mod.sampling(
        graphs, lambda b: stub_scores(b, T, S), LOOP_STEPS, sched, sched, sched, sched, torch.device("cpu"), t_to_sigma, margs,
        batch_size=2, temp_sampling=list(TEMP_SAMPLING), temp_psi=list(TEMP_PSI), temp_sigma_data=TEMP_SIGMA_DATA,
        return_full_trajectory=True)
    res = {"seed": LOOP_SEED, "steps": LOOP_STEPS, "lig_start": starts_l, "atom_start": starts_a,
           "lig_traj": torch.from_numpy(np.stack(traj)).float(),     # ligand poses at the START of each step
           "lig_out": torch.stack([d["ligand"].pos.float() for d in out]), "atom_out": torch.stack([d["atom"].pos.float() for d in out])}
    torch.save(res, OUT_LOOP)
    print("wrote", OUT_LOOP, "final ligand centre", res["lig_out"].mean((0, 1)).tolist())


if __name__ == "__main__":
    main()
    main_loop()
