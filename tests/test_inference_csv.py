"""The csv batch driver (diffdock_pocket_amd/inference.py; reference inference.py:459-493 + datasets/pdbbind.py:1005-1066) on
the CPU with a stub score function: row cleaning, skipped rows, ranking, and bitwise shard invariance of the sample-sharded
2-rank run (gloo) against the single-process run."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from diffdock_pocket_amd import inference as INF

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Stub:
    """Deterministic stand-in for the score model (pure function of the batch positions); CPU only."""
    flexible_sidechains = True

    def __call__(self, b):
        B = b.num_graphs
        lp = b["ligand"].pos.reshape(B, -1, 3)
        c = lp.mean(1)
        tr = -0.05 * c
        rot = 0.02 * torch.stack([c[:, 1], -c[:, 0], c[:, 2]], 1)
        T = int(b["ligand"].edge_mask.sum())
        tor = 0.01 * lp[:, :1, 0].expand(B, T // B).reshape(-1) if T else torch.empty(0)
        S = b["flexResidues"].edge_idx.shape[0] if len(b["flexResidues"]) > 0 else 0
        sc = 0.01 * torch.ones(S)
        return tr, rot, tor, sc


class StubConfidence:
    def __call__(self, b):
        B = b.num_graphs
        return -b["ligand"].pos.reshape(B, -1, 3).mean(1).norm(dim=1)       # closer to the pocket centre = more confident


def _write_csv(tmp_path):
    p = tmp_path / "complexes.csv"
    p.write_text(
        "complex_name,experimental_protein,ligand,pocket_center_x,pocket_center_y,pocket_center_z,flexible_sidechains\n"
        "3dpf_flex,3dpf_protein.pdb,3dpf_ligand.sdf,,,,A:160-A:193-A:197\n"
        "3dpf_smiles,3dpf_protein.pdb,COc(cc1)ccc1C#N\n"
        "3dpf_noligand,3dpf_protein.pdb,\n"
        "3dpf_rigid,3dpf_protein.pdb,3dpf_ligand.sdf\n")
    return str(p)


def _run(csv_path, rank=0, world=1, dist=None, shard="samples"):
    return INF.run_csv(csv_path, Stub(), torch.device("cpu"), confidence_model=StubConfidence(), samples_per_complex=5,
                       inference_steps=3, root=GOLDEN, seed=2, rank=rank, world=world, dist=dist, shard=shard, allow_zero_esm=True)


def test_rows_are_cleaned_like_the_reference_loader(tmp_path):
    rows = INF.load_protein_ligand_csv(_write_csv(tmp_path))
    assert [r["complex_name"] for r in rows] == ["3dpf_flex", "3dpf_smiles", "3dpf_rigid"]      # the row without a ligand is dropped
    assert rows[0]["flexible_sidechains"] == "A:160-A:193-A:197" and rows[0]["pocket_center"] is None
    assert rows[2]["flexible_sidechains"] is None


def test_csv_run_ranks_poses_and_skips_unreadable_rows(tmp_path):
    res = _run(_write_csv(tmp_path))
    assert [r.name for r in res] == ["3dpf_flex", "3dpf_smiles", "3dpf_rigid"]
    assert res[1].skipped is not None and "SDF" in res[1].skipped and res[1].ligand_pos is None
    for r in (res[0], res[2]):
        assert r.skipped is None and r.ligand_pos.shape[0] == 5 and torch.isfinite(r.ligand_pos).all()
        assert sorted(r.order.tolist()) == [0, 1, 2, 3, 4]
        assert bool((r.confidence[:-1] >= r.confidence[1:]).all())        # best first


def test_rows_without_an_esm_embedding_are_skipped_unless_asked_for(tmp_path):
    """A missing ESM embedding is not silently replaced by zeros: the row is reported as skipped with the reason."""
    csv_path = _write_csv(tmp_path)
    res = INF.run_csv(csv_path, Stub(), torch.device("cpu"), samples_per_complex=2, inference_steps=1, root=GOLDEN, seed=2)
    assert all(r.skipped is not None and r.ligand_pos is None for r in res)
    assert "no ESM embedding" in res[0].skipped and "SDF" in res[1].skipped
    with pytest.warns(RuntimeWarning, match="ZERO language-model block"):
        INF.build_row_graph(INF.load_protein_ligand_csv(csv_path)[0], root=GOLDEN, allow_zero_esm=True)


def test_full_structure_esm_embeddings_are_sliced_to_the_kept_residues(tmp_path):
    """ESM files cover the whole chain; the reference slices them with its kept-residue mask (process_mols.py:389-397).  Per-chain
    arrays, one concatenated array and the exact kept rows must give the same receptor block."""
    from diffdock_pocket_amd import inputs as I
    pdb, sdf = open(os.path.join(GOLDEN, "3dpf_protein.pdb")).read(), open(os.path.join(GOLDEN, "3dpf_ligand.sdf")).read()
    res = I.parse_pdb(pdb)
    g0 = I.build_complex_graph(pdb, sdf)
    mol = I.parse_sdf(sdf)
    import numpy as np
    centre, radius = I.binding_pocket(np.array([r.atom("CA").coord for r in res if r.atom("CA") is not None], dtype=np.float32),
                                      I.ligand_graph(mol)[1], 5.0, 0.0)
    rec = I.extract_receptor(res, I.ligand_graph(mol)[1].astype(np.float64), pocket=(centre, radius + 10.0))
    n_kept, total = len(rec.residues), sum(rec.chain_lengths)
    assert n_kept == g0["receptor"].x.shape[0] and total > n_kept and len(rec.lm_index) == n_kept
    gen = torch.Generator().manual_seed(0)
    chains = [torch.randn(n, 1280, generator=gen) for n in rec.chain_lengths]
    g_chain = I.build_complex_graph(pdb, sdf, lm_embeddings=chains)
    g_full = I.build_complex_graph(pdb, sdf, lm_embeddings=torch.cat(chains, 0))
    want = torch.stack([chains[c][i] for c, i in rec.lm_index])
    g_kept = I.build_complex_graph(pdb, sdf, lm_embeddings=want)
    for g in (g_chain, g_full, g_kept):
        assert g["receptor"].x.shape == (n_kept, 1281) and torch.equal(g["receptor"].x[:, 1:], want)
    with pytest.raises(ValueError):
        I.build_complex_graph(pdb, sdf, lm_embeddings=torch.zeros(n_kept + 1, 1280))
    # through the csv driver: a full-structure file is accepted
    csv_path = _write_csv(tmp_path)
    row = INF.load_protein_ligand_csv(csv_path)[2]
    assert row["complex_name"] == "3dpf_rigid"
    g = INF.build_row_graph(row, {"3dpf_rigid": torch.cat(chains, 0)}, GOLDEN)
    assert torch.equal(g["receptor"].x[:, 1:], want)


def _worker(rank, world, csv_path, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _run(csv_path, rank, world, dist)
    if rank == 0:
        q.put([(r.name, r.skipped, r.ligand_pos, r.confidence) for r in res])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_sharding_equals_the_single_process_run(tmp_path):
    csv_path = _write_csv(tmp_path)
    full = _run(csv_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, csv_path, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for (name, skipped, lig, conf), ref in zip(got, full):
        assert name == ref.name and (skipped is None) == (ref.skipped is None)
        if ref.ligand_pos is not None:
            assert torch.equal(lig, ref.ligand_pos) and torch.equal(conf, ref.confidence)


class FailingStub(Stub):
    """Raises on rank `bad_rank` while sampling the complex with `bad_T` rotatable bonds per graph (a stand-in for a rank-local
    DdpError: each rank holds other poses, so e.g. a truncated ligand<-atom list is seen by one rank only)."""

    def __init__(self, rank, bad_rank, fail_flex):
        self.rank, self.bad_rank, self.fail_flex = rank, bad_rank, fail_flex

    def __call__(self, b):
        has_flex = len(b["flexResidues"]) > 0
        if self.rank == self.bad_rank and has_flex == self.fail_flex:
            raise RuntimeError("injected sampling failure")
        return super().__call__(b)


def _worker_failing(rank, world, csv_path, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = INF.run_csv(csv_path, FailingStub(rank, 1, True), torch.device("cpu"), confidence_model=StubConfidence(), samples_per_complex=5,
                      inference_steps=3, root=GOLDEN, seed=2, rank=rank, world=world, dist=dist, allow_zero_esm=True)
    q.put((rank, [(r.name, r.skipped, r.ligand_pos, r.confidence) for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_a_sampling_failure_on_one_rank_skips_the_complex_on_all_ranks(tmp_path):
    """a sampling-time exception is rank-local under sample sharding.  The complex is skipped on EVERY rank (the
    reference skips a failing complex, inference.py:282-287), nobody is left waiting in the gathers, and the next complex still
    comes out bitwise as in the single-process run."""
    csv_path = _write_csv(tmp_path)
    full = _run(csv_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker_failing, args=(r, 2, csv_path, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        by_name = {name: (skipped, lig, conf) for name, skipped, lig, conf in got[rank]}
        assert by_name["3dpf_flex"][0] is not None and by_name["3dpf_flex"][1] is None
        assert ("injected" in by_name["3dpf_flex"][0]) == (rank == 1) and ("another rank" in by_name["3dpf_flex"][0]) == (rank == 0)
        ref = [r for r in full if r.name == "3dpf_rigid"][0]
        assert by_name["3dpf_rigid"][0] is None and torch.equal(by_name["3dpf_rigid"][1], ref.ligand_pos)
        assert torch.equal(by_name["3dpf_rigid"][2], ref.confidence)
    # single process: the failing complex is reported, the run goes on
    res = INF.run_csv(csv_path, FailingStub(0, 0, True), torch.device("cpu"), samples_per_complex=2, inference_steps=1, root=GOLDEN, seed=2,
                      allow_zero_esm=True)
    assert "injected" in res[0].skipped and res[2].skipped is None and res[2].ligand_pos is not None


def test_complex_sharding_follows_array_split(tmp_path):
    csv_path = _write_csv(tmp_path)
    r0, r1 = _run(csv_path, 0, 2, shard="complexes"), _run(csv_path, 1, 2, shard="complexes")
    done0 = [r.name for r in r0 if r.ligand_pos is not None or r.skipped]
    done1 = [r.name for r in r1 if r.ligand_pos is not None or r.skipped]
    assert done0 == ["3dpf_flex", "3dpf_smiles"] and done1 == ["3dpf_rigid"]       # np.array_split([0, 1, 2], 2)
