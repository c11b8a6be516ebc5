"""Input pipeline (diffdock_pocket_amd/inputs.py) against vectors produced by the reference's own
functions (oracle/make_golden_inputs.py -> tests/golden/inputs_3dpf.npz) on the reference's example complex, and
against the geometry fixture the benchmark uses (assets/3dpf_geometry.npz, made with networkx by
oracle/make_3dpf_geometry.py)."""
import os

import numpy as np
import pytest
import torch

from diffdock_pocket_amd import inputs as I

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FLEX = "A:160-A:193-A:197-A:198-A:222-A:224-A:227"


@pytest.fixture(scope="module")
def gold():
    with np.load(os.path.join(GOLD, "inputs_3dpf.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def graph():
    pdb = open(os.path.join(GOLD, "3dpf_protein.pdb")).read()
    sdf = open(os.path.join(GOLD, "3dpf_ligand.sdf")).read()
    return I.build_complex_graph(pdb, sdf, name="3dpf", flexible_sidechains=FLEX)


def test_ligand_graph_matches_reference(graph, gold):
    c = graph.original_center.numpy()
    assert np.allclose(graph["ligand"].pos.numpy() + c, gold["lig_pos"], atol=1e-5)
    assert np.array_equal(graph["ligand", "ligand"].edge_index.numpy(), gold["lig_edge_index"])
    assert np.array_equal(graph["ligand", "ligand"].edge_attr.numpy(), gold["lig_edge_attr"])
    assert graph["ligand"].x.shape == (37, 16) and graph["ligand"].x.dtype == torch.long


def test_feature_index_mapping_matches_reference(gold):
    """lig_atom_featurizer's index mapping on given atom properties (incl. out-of-vocabulary values -> 'misc')."""
    probes = [eval(s) for s in gold["probe_table"]]        # dict literals written by the generator
    props = {"atomic_num": [p["z"] for p in probes], "chirality": [p["chi"] for p in probes],
             "degree": [p["deg"] for p in probes], "formal_charge": [p["fc"] for p in probes],
             "implicit_valence": [p["iv"] for p in probes], "num_h": [p["nh"] for p in probes],
             "radical_e": [p["rad"] for p in probes], "hybridization": [p["hyb"] for p in probes],
             "aromatic": [p["arom"] for p in probes], "num_rings": [p["nring"] for p in probes],
             "ring_sizes": [set(p["sizes"]) for p in probes]}
    assert np.array_equal(I.lig_atom_features(props).numpy(), gold["probe_feats"])


def test_pocket_and_receptor_match_reference(graph, gold):
    c = graph.original_center.numpy()
    assert np.allclose(c[0], gold["pocket_centre"], atol=1e-5)
    assert np.array_equal(graph["receptor"].x.numpy(), gold["rec_x"])
    assert np.allclose(graph["receptor"].pos.numpy() + c, gold["rec_pos"], atol=1e-4)
    assert np.array_equal(graph["receptor", "receptor"].edge_index.numpy(), gold["rec_edge_index"])
    assert np.array_equal(graph["atom"].x.numpy(), gold["atom_x"])
    assert np.allclose(graph["atom"].pos.numpy() + c, gold["atom_pos"], atol=1e-4)
    assert np.array_equal(graph["atom", "receptor"].edge_index.numpy(), gold["atom_res"])


def test_sidechain_masks_match_reference(graph, gold):
    fr = graph["flexResidues"]
    assert np.array_equal(fr.edge_idx.numpy(), gold["flex_edge_idx"])
    assert np.array_equal(fr.subcomponentsMapping.numpy(), gold["flex_mapping"])
    # the rotated atoms of a bond form a set upstream (python set iteration order): compare per bond as sets
    for (a, b), (ga, gb) in zip(fr.subcomponentsMapping.tolist(), gold["flex_mapping"].tolist()):
        assert set(fr.subcomponents[a:b].tolist()) == set(gold["flex_subcomponents"][ga:gb].tolist())
    assert np.array_equal(fr.residueNBondsMapping.numpy(), gold["flex_n_bonds"])
    assert [f"{c}:{i}" for c, i in fr.pdbIds] == list(gold["flex_ids"])


def test_rotatable_bond_masks_match_networkx_fixture(graph):
    from diffdock_pocket_amd.synthetic import load_3dpf_geometry
    g = load_3dpf_geometry()
    assert np.array_equal(graph["ligand"].edge_mask.numpy(), g["lig_edge_mask"])
    assert np.array_equal(graph["ligand"].mask_rotate, g["lig_mask_rotate"])


def test_perception_on_known_molecules():
    """Chemistry perception (parity unpinned vs rdkit, see inputs.py): textbook cases must come out right."""
    def mol(elements, bonds, charges=None):
        return I.Molecule(np.zeros((len(elements), 3)), elements, bonds, charges or [0] * len(elements))

    # benzene, kekulised, hydrogens implicit (1 each)
    benz = mol(["C"] * 6, [(i, (i + 1) % 6, 2 if i % 2 == 0 else 1) for i in range(6)])
    p = I.perceive(benz, [1] * 6)
    assert all(p["aromatic"]) and p["hybridization"] == ["SP2"] * 6 and p["num_rings"] == [1] * 6
    assert p["ring_sizes"] == [{6}] * 6 and p["degree"] == [3] * 6
    # cyclohexane: not aromatic, sp3
    p = I.perceive(mol(["C"] * 6, [(i, (i + 1) % 6, 1) for i in range(6)]), [2] * 6)
    assert not any(p["aromatic"]) and p["hybridization"] == ["SP3"] * 6
    # pyrrole: N-H lone pair completes the sextet
    pyr = mol(["N", "C", "C", "C", "C"], [(0, 1, 1), (1, 2, 2), (2, 3, 1), (3, 4, 2), (4, 0, 1)])
    p = I.perceive(pyr, [1, 1, 1, 1, 1])
    assert all(p["aromatic"]) and p["hybridization"][0] == "SP2"
    # acetamide C-C(=O)-N: carbonyl C sp2, amide N conjugated -> sp2, methyl sp3; acetonitrile C#N: sp
    p = I.perceive(mol(["C", "C", "O", "N"], [(0, 1, 1), (1, 2, 2), (1, 3, 1)]), [3, 0, 0, 2])
    assert p["hybridization"] == ["SP3", "SP2", "SP2", "SP2"] and not any(p["aromatic"])
    p = I.perceive(mol(["C", "C", "N"], [(0, 1, 1), (1, 2, 3)]), [3, 0, 0])
    assert p["hybridization"] == ["SP3", "SP", "SP"]
    # naphthalene (kekulised): two fused six-rings, the fusion atoms sit in both
    bonds = [(0, 1, 2), (1, 2, 1), (2, 3, 2), (3, 4, 1), (4, 5, 2), (5, 0, 1), (4, 6, 1), (6, 7, 2), (7, 8, 1), (8, 9, 2), (9, 5, 1)]
    p = I.perceive(mol(["C"] * 10, bonds), [1, 1, 1, 1, 0, 0, 1, 1, 1, 1])
    assert all(p["aromatic"]) and p["num_rings"][4] == 2 and p["num_rings"][0] == 1
    # spiro / bridged ring counts: cubane-like check of the ring-basis size (E - V + 1)
    cube = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
    assert len(I._smallest_rings(8, cube)) == 5 and all(len(r) == 4 for r in I._smallest_rings(8, cube))


def test_sdf_charges_and_pdb_altloc():
    sdf = "\n".join(["m", "", "", "  2  1  0  0  0  0  0  0  0  0999 V2000",
                     "    0.0000    0.0000    0.0000 N   0  3  0  0  0", "    1.0000    0.0000    0.0000 O   0  0  0  0  0",
                     "  1  2  1  0", "M  CHG  1   2  -1", "M  END", "$$$$"])
    m = I.parse_sdf(sdf)
    assert m.charges == [0, -1] and m.bonds == [(0, 1, 1)]          # the property block replaces the atom-block code
    pdb = "\n".join([
        "ATOM      1  N   ALA A   1      11.104  13.207   2.100  1.00  0.00           N",
        "ATOM      2  CA AALA A   1      12.000  13.000   2.000  0.40  0.00           C",
        "ATOM      3  CA BALA A   1      12.500  13.000   2.000  0.60  0.00           C",
        "ATOM      4  C   ALA A   1      13.000  14.000   2.000  1.00  0.00           C",
        "HETATM    5  O   HOH A 101      20.000  20.000  20.000  1.00  0.00           O",
        "ENDMDL", "ATOM      6  N   ALA A   2      11.104  13.207   2.100  1.00  0.00           N"])
    res = I.parse_pdb(pdb)
    assert [r.resname for r in res] == ["ALA", "HOH"] and res[1].hetflag == "W"
    assert len(res[0].atoms) == 3 and abs(float(res[0].atom("CA").coord[0]) - 12.5) < 1e-6


def test_graph_feeds_collate_and_schema(graph):
    """The graph has the fields the score model reads and survives the batch container."""
    from diffdock_pocket_amd.batch import collate, set_time
    b = collate([graph, graph])
    set_time(b, 0.5, 0.5, 0.5, 0.5)
    assert b.num_graphs == 2 and b["ligand"].x.shape[0] == 74 and b["atom", "receptor"].edge_index.shape[1] == 2 * graph["atom"].x.shape[0]
    assert int(b["atom", "receptor"].edge_index[1].max()) == 2 * graph["receptor"].x.shape[0] - 1
    assert b["flexResidues"].edge_idx.shape == (16, 2)


def test_receptor_filters_chains_waters_and_incomplete_residues():
    """extract_receptor (reference datasets/process_mols.py:291-432): waters and residues without N / CA / C are dropped,
    a chain far from the ligand is dropped, modified residues in HETATM records with a backbone are kept, insertion codes
    make distinct residues."""
    def atom(i, name, res, chain, seq, x, rec="ATOM  ", icode=" ", el=None):
        return f"{rec}{i:5d} {name:<4s} {res:>3s} {chain}{seq:4d}{icode}   {x:8.3f}{0.0:8.3f}{0.0:8.3f}  1.00  0.00          {el or name[0]:>2s}"

    lines, i = [], 1
    def residue(res, chain, seq, x0, rec="ATOM  ", icode=" ", names=("N", "CA", "C", "O", "CB")):
        nonlocal i
        for k, n in enumerate(names):
            lines.append(atom(i, n, res, chain, seq, x0 + 0.5 * k, rec, icode))
            i += 1
    residue("ALA", "A", 1, 0.0)
    residue("SER", "A", 1, 3.0, icode="A")                       # insertion code: its own residue
    residue("MSE", "A", 2, 6.0, rec="HETATM")                    # modified residue with a backbone: kept
    residue("GLY", "A", 3, 9.0, names=("N", "CA"))               # no C: dropped
    lines.append(atom(i, "O", "HOH", "A", 100, 1.0, "HETATM")); i += 1
    residue("LEU", "B", 1, 200.0)                                # chain B is 200 A away
    residue("VAL", "B", 2, 203.0)
    res = I.parse_pdb("\n".join(lines))
    lig = np.array([[1.0, 1.0, 0.0], [2.0, 1.0, 0.0]])
    rec = I.extract_receptor(res, lig, cutoff=10.0)
    assert [(r.chain, r.resseq, r.icode, r.resname) for r in rec.residues] == [("A", 1, " ", "ALA"), ("A", 1, "A", "SER"), ("A", 2, " ", "MSE")]
    assert rec.ca.shape == (3, 3) and I.rec_residue_features(rec)[:, 0].tolist() == [0.0, 15.0, 37.0]    # MSE -> 'misc'
    rr, ax, apos, ares, heavy = I.receptor_graph(rec, cutoff=15.0, max_neighbors=24)
    assert ax.shape == (15, 4) and ares.tolist() == [0] * 5 + [1] * 5 + [2] * 5 and sorted(set(rr[0].tolist())) == [0, 1, 2]
    # nothing within the chain cutoff: the closest chain is kept (process_mols.py:388-389)
    far = I.extract_receptor(res, lig + 1000.0, cutoff=10.0)
    assert {r.chain for r in far.residues} == {"B"}
