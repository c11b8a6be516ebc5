"""Input pipeline without rdkit / biopython: PDB + SDF text -> the complex graph the score model consumes.

Counterpart of the reference's data layer for ONE complex at inference time, restricted to what the hot path reads:

  parse_pdb / parse_sdf                    plain-text readers (Bio.PDB.PDBParser / rdkit MolFromMolFile upstream)
  ligand_graph                             reference datasets/process_mols.py:435-453 (`get_lig_graph`) + :115-140 featuriser
  transformation_mask                      reference utils/torsion.py:16-65 (`get_transformation_mask`)
  extract_receptor                         reference datasets/process_mols.py:291-432 (`extract_receptor_structure`)
  binding_pocket / pocket selector         reference datasets/pdbbind.py:324-339, :585-603, :775-784 (mode 'center-dist')
  receptor_graph                           reference datasets/process_mols.py:650-723 (`get_fullrec_graph`)
  sidechain_rotation_masks                 reference datasets/process_mols.py:773-883 + utils/torsion.py:165-248
  build_complex_graph                      the order of reference datasets/pdbbind.py `get_complex` incl. centring (:704-731)

Everything that is arithmetic on coordinates, names and indices follows the reference line by line and is pinned by
`tests/test_inputs.py` against fixtures produced with the reference's own functions (oracle/make_golden_inputs.py) on the
reference's example complex.  What rdkit PERCEIVES about a molecule is not reproducible without rdkit and is restated
from its published rules - **parity unpinned** for: aromaticity (Hueckel count on the smallest rings), hybridisation
(bonds + lone pairs, conjugated N/O/S -> SP2), ring membership (smallest set of smallest rings via shortest cycles) and
chirality tags (always CHI_UNSPECIFIED).  These only feed embedding-table indices of the ligand node encoder.

Host-side Python / numpy by design (north_star: graph construction stays on the host); nothing here is on the timed path.
"""
from __future__ import annotations

import re
from collections import deque
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .batch import HeteroBatch, Store

# ---------------------------------------------------------------------------------------------- vocabularies
# reference datasets/process_mols.py:32-63 (`allowable_features`), restated as data
ATOMIC_NUMS = list(range(1, 119)) + ["misc"]
CHIRALITY = ["CHI_UNSPECIFIED", "CHI_TETRAHEDRAL_CW", "CHI_TETRAHEDRAL_CCW", "CHI_OTHER"]
DEGREES = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, "misc"]
NUMRING = [0, 1, 2, 3, 4, 5, 6, "misc"]
IMPLICIT_VALENCE = [0, 1, 2, 3, 4, 5, 6, "misc"]
FORMAL_CHARGE = [-5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5, "misc"]
NUM_H = [0, 1, 2, 3, 4, 5, 6, 7, 8, "misc"]
RADICAL_E = [0, 1, 2, 3, 4, "misc"]
HYBRIDIZATION = ["SP", "SP2", "SP3", "SP3D", "SP3D2", "misc"]
AMINO_ACIDS = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO", "SER",
               "THR", "TRP", "TYR", "VAL", "HIP", "HIE", "TPO", "HID", "LEV", "MEU", "PTR", "GLV", "CYT", "SEP", "HIZ", "CYM",
               "GLM", "ASQ", "TYS", "CYX", "GLZ", "misc"]
ATOM_TYPE_2 = ["C*", "CA", "CB", "CD", "CE", "CG", "CH", "CZ", "N*", "ND", "NE", "NH", "NZ", "O*", "OD", "OE", "OG", "OH", "OX",
               "S*", "SD", "SG", "misc"]
ATOM_TYPE_3 = ["C", "CA", "CB", "CD", "CD1", "CD2", "CE", "CE1", "CE2", "CE3", "CG", "CG1", "CG2", "CH2", "CZ", "CZ2", "CZ3", "N",
               "ND1", "ND2", "NE", "NE1", "NE2", "NH1", "NH2", "NZ", "O", "OD1", "OD2", "OE1", "OE2", "OG", "OG1", "OH", "OXT",
               "SD", "SG", "misc"]
FLEXIBLE_SIDECHAINS = {"ARG", "HIS", "LYS", "ASP", "GLU", "SER", "THR", "ASN", "GLN", "CYS", "SEC", "GLY", "PRO", "ALA", "VAL",
                       "ILE", "LEU", "MET", "PHE", "TYR", "TRP"}
BOND_TYPES = {1: 0, 2: 1, 3: 2, 4: 3}          # SDF bond order -> reference `bonds` index (process_mols.py:67)

_SYMBOLS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo "
            "Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl "
            "Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv Ts Og").split()
ATOMIC_NUMBER = {s.upper(): i + 1 for i, s in enumerate(_SYMBOLS)}
_OUTER_ELECTRONS = {1: 1, 5: 3, 6: 4, 7: 5, 8: 6, 9: 7, 14: 4, 15: 5, 16: 6, 17: 7, 34: 6, 35: 7, 53: 7}


def safe_index(lst, e):
    """reference datasets/process_mols.py:165-170: index of e, last index ('misc') when absent."""
    try:
        return lst.index(e)
    except ValueError:
        return len(lst) - 1


# ---------------------------------------------------------------------------------------------- text readers
@dataclass
class PdbAtom:
    name: str
    element: str
    coord: np.ndarray      # float32[3] like Bio.PDB atoms
    occupancy: float = 1.0


@dataclass
class PdbResidue:
    chain: str
    hetflag: str           # ' ' for ATOM records, 'W' water, 'H_xxx' other HETATM (Bio.PDB residue id[0])
    resseq: int
    icode: str
    resname: str
    atoms: List[PdbAtom] = field(default_factory=list)

    def atom(self, name) -> Optional[PdbAtom]:
        for a in self.atoms:
            if a.name == name:
                return a
        return None


def _element_from_name(fullname: str) -> str:
    """Bio.PDB's fallback when columns 77-78 are empty: leading alphabetic character(s) of the atom name."""
    name = fullname.strip()
    if fullname[:1].isalpha() and not fullname[:2].strip().isdigit() and len(name) >= 2 and fullname[0] != " " \
            and name[:2].upper() in ATOMIC_NUMBER and not name[1].isdigit() and len(fullname.strip()) < 4:
        return name[:2].upper()
    for ch in name:
        if ch.isalpha():
            return ch.upper()
    return ""


def parse_pdb(text: str) -> List[PdbResidue]:
    """ATOM / HETATM records of the first MODEL, grouped into residues in file order (chain by chain as they first
    appear, like iterating a Bio.PDB structure).  Of alternate locations the one with the highest occupancy is kept (first
    wins on ties) - Bio.PDB's DisorderedAtom selection."""
    chains: Dict[str, Dict[Tuple, PdbResidue]] = {}
    alt_best: Dict[Tuple, float] = {}
    for ln in text.splitlines():
        rec = ln[:6]
        if rec.startswith("ENDMDL"):
            break
        if rec not in ("ATOM  ", "HETATM"):
            continue
        fullname, altloc, resname, chain = ln[12:16], ln[16], ln[17:20].strip(), ln[21]
        resseq, icode = int(ln[22:26]), ln[26]
        coord = np.array([float(ln[30:38]), float(ln[38:46]), float(ln[46:54])], dtype=np.float32)
        try:
            occ = float(ln[54:60])
        except ValueError:
            occ = 1.0
        element = ln[76:78].strip().upper() if len(ln) >= 78 else ""
        if not element:
            element = _element_from_name(fullname)
        hetflag = " " if rec == "ATOM  " else ("W" if resname in ("HOH", "WAT") else "H_" + resname)
        key = (hetflag, resseq, icode)
        res = chains.setdefault(chain, {}).get(key)
        if res is None:
            res = chains[chain][key] = PdbResidue(chain, hetflag, resseq, icode, resname)
        name = fullname.strip()
        akey = (chain, key, name)
        old = res.atom(name)
        if old is not None:
            if altloc != " " and occ > alt_best.get(akey, old.occupancy):
                old.coord, old.occupancy, old.element = coord, occ, element
                alt_best[akey] = occ
            continue
        res.atoms.append(PdbAtom(name, element, coord, occ))
        alt_best[akey] = occ
    return [r for ch in chains.values() for r in ch.values()]


@dataclass
class Molecule:
    pos: np.ndarray                 # float64 [n,3]
    elements: List[str]
    bonds: List[Tuple[int, int, int]]     # (a, b, SDF bond type 1/2/3/4)
    charges: List[int]


def parse_sdf(text: str) -> Molecule:
    """First record of an SDF / MOL file, V2000 connection table (atom block, bond block, `M  CHG` lines)."""
    lines = text.splitlines()
    if len(lines) < 4 or "V2000" not in lines[3]:
        raise ValueError("only V2000 MOL/SDF records are supported")
    na, nb = int(lines[3][0:3]), int(lines[3][3:6])
    pos, elem, chg = [], [], []
    old_code = {0: 0, 1: 3, 2: 2, 3: 1, 4: 0, 5: -1, 6: -2, 7: -3}
    for ln in lines[4:4 + na]:
        pos.append([float(ln[0:10]), float(ln[10:20]), float(ln[20:30])])
        elem.append(ln[31:34].strip())
        code = ln[36:39].strip()
        chg.append(old_code.get(int(code), 0) if code else 0)
    bonds = []
    for ln in lines[4 + na:4 + na + nb]:
        bonds.append((int(ln[0:3]) - 1, int(ln[3:6]) - 1, int(ln[6:9])))
    m_chg = False
    for ln in lines[4 + na + nb:]:
        if ln.startswith("M  END") or ln.startswith("$$$$"):
            break
        if ln.startswith("M  CHG"):
            if not m_chg:            # the property block supersedes the atom-block codes
                chg, m_chg = [0] * na, True
            f = ln[6:].split()
            for a, c in zip(f[1::2], f[2::2]):
                chg[int(a) - 1] = int(c)
    return Molecule(np.array(pos, np.float64).reshape(-1, 3), elem, bonds, chg)


# ---------------------------------------------------------------------------------------------- small graph helpers
def _components(n: int, adj: Sequence[Iterable[int]]) -> List[List[int]]:
    """Connected components in order of their lowest node, nodes in BFS discovery order (networkx iteration order)."""
    seen, out = [False] * n, []
    for s in range(n):
        if seen[s]:
            continue
        seen[s] = True
        comp, q = [s], deque([s])
        while q:
            u = q.popleft()
            for v in adj[u]:
                if not seen[v]:
                    seen[v] = True
                    comp.append(v)
                    q.append(v)
        out.append(comp)
    return out


def _smallest_rings(n: int, bonds: Sequence[Tuple[int, int]]) -> List[List[int]]:
    """A smallest set of smallest rings: shortest cycle through every ring bond, then a linearly independent subset
    (GF(2) over bonds) in order of size, E - V + C rings in total."""
    adj = [[] for _ in range(n)]
    for a, b in bonds:
        adj[a].append(b)
        adj[b].append(a)
    n_rings = len(bonds) - n + len(_components(n, adj))
    if n_rings <= 0:
        return []
    bond_id = {frozenset(b): i for i, b in enumerate(bonds)}
    cand = {}
    for a, b in bonds:
        prev, q = {a: -1}, deque([a])          # shortest path a -> b that avoids the bond itself
        while q and b not in prev:
            u = q.popleft()
            for v in adj[u]:
                if v in prev or (u == a and v == b):
                    continue
                prev[v] = u
                q.append(v)
        if b not in prev:
            continue
        path, u = [], b
        while u != -1:
            path.append(u)
            u = prev[u]
        cand.setdefault(frozenset(path), path)
    rings, basis = [], []
    for path in sorted(cand.values(), key=len):
        vec = 0
        for i in range(len(path)):
            vec ^= 1 << bond_id[frozenset((path[i], path[(i + 1) % len(path)]))]
        for bvec in basis:
            vec = min(vec, vec ^ bvec)
        if vec:
            basis.append(vec)
            rings.append(path)
            if len(rings) == n_rings:
                break
    return rings


# ---------------------------------------------------------------------------------------------- ligand
def remove_hs(mol: Molecule) -> Tuple[Molecule, List[int]]:
    """Heavy-atom molecule + number of hydrogens that were attached to each kept atom (rdkit RemoveHs: the removed
    hydrogens turn into implicit ones)."""
    heavy = [i for i, e in enumerate(mol.elements) if e.upper() != "H"]
    remap = {a: k for k, a in enumerate(heavy)}
    nh = [0] * len(heavy)
    bonds = []
    for a, b, o in mol.bonds:
        if a in remap and b in remap:
            bonds.append((remap[a], remap[b], o))
        elif a in remap:
            nh[remap[a]] += 1
        elif b in remap:
            nh[remap[b]] += 1
    return Molecule(mol.pos[heavy], [mol.elements[i] for i in heavy], bonds, [mol.charges[i] for i in heavy]), nh


def perceive(mol: Molecule, num_h: Sequence[int]) -> Dict[str, list]:
    """Per-atom properties that rdkit would report (see the module docstring for what is approximated)."""
    n = len(mol.elements)
    z = [ATOMIC_NUMBER.get(e.upper(), -1) for e in mol.elements]
    nbrs = [[] for _ in range(n)]
    for a, b, o in mol.bonds:
        nbrs[a].append((b, o))
        nbrs[b].append((a, o))
    rings = _smallest_rings(n, [(a, b) for a, b, _ in mol.bonds])
    ring_sets = [set(r) for r in rings]

    def lone_pairs(i, valence):
        outer = _OUTER_ELECTRONS.get(z[i])
        return 0 if outer is None else max(0, (outer - mol.charges[i] - valence) // 2)

    aromatic = [any(o == 4 for _, o in nbrs[i]) for i in range(n)]
    if not any(aromatic):   # kekulised input: Hueckel count on each smallest ring of 5..7 atoms, repeated so that rings
        changed = True      # fused to an aromatic ring can use its (delocalised) bonds
        while changed:
            changed = False
            for r, rs in zip(rings, ring_sets):
                if not 5 <= len(r) <= 7 or all(aromatic[i] for i in r):
                    continue
                electrons, ok = 0, True
                for i in r:
                    dbl_in = any(o == 2 and j in rs for j, o in nbrs[i])
                    dbl_out = [j for j, o in nbrs[i] if o == 2 and j not in rs]
                    val = int(sum(o if o < 4 else 1.5 for _, o in nbrs[i])) + num_h[i]
                    if dbl_in or (aromatic[i] and not dbl_out):
                        electrons += 1
                    elif dbl_out:
                        ok = z[dbl_out[0]] in (8, 16, 7) and z[i] == 6      # C=O / C=S / C=N outside the ring: 0 electrons
                    elif z[i] in (7, 8, 16) and lone_pairs(i, val) > 0:
                        electrons += 2
                    elif z[i] == 6 and mol.charges[i] == -1:
                        electrons += 2
                    elif z[i] in (6, 5) and mol.charges[i] == 1:
                        electrons += 0
                    else:
                        ok = False
                    if not ok:
                        break
                if ok and electrons % 4 == 2:
                    for i in r:
                        if not aromatic[i]:
                            aromatic[i], changed = True, True

    hybrid = []
    for i in range(n):
        order = sum(o if o < 4 else 1.5 for _, o in nbrs[i])
        val = int(order + 0.5) + num_h[i]
        deg = len(nbrs[i]) + num_h[i]
        lp = lone_pairs(i, val)
        norbs = deg + lp
        h = {0: "S", 1: "S", 2: "SP", 3: "SP2", 4: "SP3", 5: "SP3D", 6: "SP3D2"}.get(norbs, "OTHER")
        if h == "SP3" and lp > 0 and z[i] in (7, 8, 16):
            # conjugation: a lone pair next to a pi system is planar (rdkit's ConjugHybrid)
            if aromatic[i] or any(aromatic[j] or any(o2 in (2, 3) for _, o2 in nbrs[j]) for j, _ in nbrs[i]):
                h = "SP2"
        if aromatic[i] and h == "SP3":
            h = "SP2"
        hybrid.append(h)
    return {
        "atomic_num": z, "degree": [len(nbrs[i]) + num_h[i] for i in range(n)], "formal_charge": list(mol.charges),
        "implicit_valence": list(num_h), "num_h": list(num_h), "radical_e": [0] * n, "hybridization": hybrid,
        "aromatic": aromatic, "num_rings": [sum(i in rs for rs in ring_sets) for i in range(n)],
        "ring_sizes": [{len(rs) for rs in ring_sets if i in rs} for i in range(n)],
        "chirality": ["CHI_UNSPECIFIED"] * n,
    }


def lig_atom_features(props: Dict[str, list]) -> torch.Tensor:
    """reference datasets/process_mols.py:115-140: the 16 categorical indices per ligand atom."""
    rows = []
    for i in range(len(props["atomic_num"])):
        rs = props["ring_sizes"][i]
        rows.append([
            safe_index(ATOMIC_NUMS, props["atomic_num"][i]),
            CHIRALITY.index(props["chirality"][i]),
            safe_index(DEGREES, props["degree"][i]),
            safe_index(FORMAL_CHARGE, props["formal_charge"][i]),
            safe_index(IMPLICIT_VALENCE, props["implicit_valence"][i]),
            safe_index(NUM_H, props["num_h"][i]),
            safe_index(RADICAL_E, props["radical_e"][i]),
            safe_index(HYBRIDIZATION, props["hybridization"][i]),
            [False, True].index(bool(props["aromatic"][i])),
            safe_index(NUMRING, props["num_rings"][i]),
        ] + [[False, True].index(k in rs) for k in (3, 4, 5, 6, 7, 8)])
    return torch.tensor(rows, dtype=torch.long).reshape(-1, 16)


def transformation_mask(n_atoms: int, edge_index: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """reference utils/torsion.py:16-65 on the directed bond list [2, 2*bonds] (consecutive pairs = one bond): a bond is
    rotatable when removing it splits the molecule and the smaller side has more than one atom; that side rotates."""
    edges = edge_index.T
    adj = [set() for _ in range(n_atoms)]
    for a, b in edges:
        adj[int(a)].add(int(b))
        adj[int(b)].add(int(a))
    to_rotate: List[List[int]] = []
    for i in range(0, edges.shape[0], 2):
        a, b = int(edges[i, 0]), int(edges[i, 1])
        assert a == int(edges[i + 1, 1])
        adj[a].discard(b)
        adj[b].discard(a)
        comps = _components(n_atoms, adj)
        adj[a].add(b)
        adj[b].add(a)
        if len(comps) > 1:
            comp = sorted(comps, key=len)[0]
            if len(comp) > 1:
                to_rotate += ([[], comp] if a in comp else [comp, []])
                continue
        to_rotate += [[], []]
    mask_edges = np.asarray([len(l) > 0 for l in to_rotate], dtype=bool)
    mask_rotate = np.zeros((int(mask_edges.sum()), n_atoms), dtype=bool)
    idx = 0
    for i in range(len(to_rotate)):
        if mask_edges[i]:
            mask_rotate[idx][np.asarray(to_rotate[i], dtype=int)] = True
            idx += 1
    return mask_edges, mask_rotate


def ligand_graph(mol: Molecule, keep_hs: bool = False):
    """(x [n,16] int64, pos [n,3] f32, edge_index [2,2b], edge_attr [2b,4], edge_mask, mask_rotate) of
    reference datasets/process_mols.py:435-453,505-512 (`get_lig_graph` + `get_transformation_mask`)."""
    if keep_hs:
        heavy, nh = mol, [0] * len(mol.elements)
    else:
        heavy, nh = remove_hs(mol)
    x = lig_atom_features(perceive(heavy, nh))
    row, col, et = [], [], []
    for a, b, o in heavy.bonds:
        row += [a, b]
        col += [b, a]
        et += 2 * [BOND_TYPES.get(o, 0)]
    edge_index = np.array([row, col], dtype=np.int64).reshape(2, -1)
    edge_attr = np.eye(4, dtype=np.float32)[np.array(et, dtype=np.int64)].reshape(-1, 4)
    edge_mask, mask_rotate = transformation_mask(len(heavy.elements), edge_index)
    return x, heavy.pos.astype(np.float32), edge_index, edge_attr, edge_mask, mask_rotate


# ---------------------------------------------------------------------------------------------- receptor
def binding_pocket(ca_coords: np.ndarray, lig_pos: np.ndarray, pocket_cutoff: float = 5.0, buffer: float = 0.0):
    """reference datasets/pdbbind.py:324-339: centre = mean of the C-alphas within pocket_cutoff of any ligand atom (the
    closest C-alpha when there is none), radius = largest ligand-atom distance from the centre + buffer."""
    d = np.linalg.norm(ca_coords[:, None, :] - lig_pos[None, :, :], axis=-1)
    label = (d < pocket_cutoff).any(1)
    centre = ca_coords[label].mean(0) if label.any() else ca_coords[d.min(1).argmin()]
    return centre, float(np.linalg.norm(lig_pos - centre[None, :], axis=1).max() + buffer)


@dataclass
class Receptor:
    residues: List[PdbResidue]         # kept residues, in structure order (what `rec.get_residues()` yields afterwards)
    ca: np.ndarray
    n: np.ndarray
    c: np.ndarray
    lm_index: Optional[List[Tuple[int, int]]] = None   # per kept residue: (chain number in the structure, index among that
                                                       # chain's amino-acid residues BEFORE the pocket selector) = its row in
                                                       # the reference's per-chain ESM embeddings (process_mols.py:346,395-397)
    chain_lengths: Optional[List[int]] = None          # amino-acid residues per chain (rows of a per-chain embedding)


def extract_receptor(residues: Sequence[PdbResidue], lig_pos: np.ndarray, cutoff: float = 10.0,
                     pocket: Optional[Tuple[np.ndarray, float]] = None) -> Receptor:
    """reference datasets/process_mols.py:291-432 without miscellaneous atoms / LM embeddings: drops waters and residues
    without CA, N and C, applies the pocket selector (pdbbind.py:775-784, all-atom rule: any atom inside the sphere), and
    keeps the chains that come within `cutoff` of the ligand (the closest chain when none does)."""
    chains: Dict[str, List[PdbResidue]] = {}
    seq_pos: Dict[int, Tuple[int, int]] = {}     # id(residue) -> (chain number, index among the chain's amino-acid residues)
    n_aa: Dict[str, int] = {}
    for r in residues:
        chains.setdefault(r.chain, [])
        n_aa.setdefault(r.chain, 0)
        if r.resname == "HOH":
            continue
        if r.atom("CA") is None or r.atom("N") is None or r.atom("C") is None:
            continue
        seq_pos[id(r)] = (list(chains).index(r.chain), n_aa[r.chain])     # counted before the selector (`atom_idx`, :346)
        n_aa[r.chain] += 1
        if pocket is not None:
            xyz = np.array([a.coord for a in r.atoms])
            if not (np.linalg.norm(xyz - pocket[0], axis=1) < pocket[1]).any():
                continue
        chains[r.chain].append(r)
    ids, dmin = list(chains), []
    for cid in ids:
        if chains[cid]:
            xyz = np.concatenate([np.array([a.coord for a in r.atoms]) for r in chains[cid]], 0)
            dmin.append(float(np.linalg.norm(lig_pos[:, None, :] - xyz[None, :, :], axis=-1).min()))
        else:
            dmin.append(np.inf)
    valid = [cid for cid, d in zip(ids, dmin) if d < cutoff] or [ids[int(np.argmin(dmin))]]
    kept = [r for cid in ids if cid in valid for r in chains[cid]]
    if not kept:
        raise ValueError("no receptor residue left")

    def vec(name):
        return np.array([r.atom(name).coord for r in kept], dtype=np.float32)

    return Receptor(kept, vec("CA"), vec("N"), vec("C"), lm_index=[seq_pos[id(r)] for r in kept],
                    chain_lengths=[n_aa[cid] for cid in ids])


def slice_lm_embeddings(lm_embeddings, rec: Receptor) -> torch.Tensor:
    """ESM rows of the kept residues, [n_kept, 1280] (see build_complex_graph for the accepted forms)."""
    def as_t(a):
        return torch.as_tensor(np.asarray(a.cpu() if torch.is_tensor(a) else a), dtype=torch.float32)

    n_kept = len(rec.residues)
    if isinstance(lm_embeddings, (list, tuple)):
        chains = [as_t(c) for c in lm_embeddings]
        rows = []
        for ci, ri in rec.lm_index:
            if ci >= len(chains):    # process_mols.py:391-392
                raise ValueError("Encountered valid chain id that was not present in the LM embeddings")
            if ri >= chains[ci].shape[0]:
                raise ValueError(f"LM embedding of chain {ci} has {chains[ci].shape[0]} rows, residue {ri} requested")
            rows.append(chains[ci][ri])
        lm = torch.stack(rows, 0)
    else:
        lm = as_t(lm_embeddings)
        total = sum(rec.chain_lengths)
        if lm.shape[0] == total and total != n_kept:
            starts = np.concatenate([[0], np.cumsum(rec.chain_lengths)[:-1]])
            lm = lm[[int(starts[ci]) + ri for ci, ri in rec.lm_index]]
        elif lm.shape[0] != n_kept:
            raise ValueError(f"lm_embeddings has {lm.shape[0]} rows; the receptor keeps {n_kept} of the structure's {total} "
                             f"amino-acid residues (give per-chain arrays, all {total} rows, or exactly the kept rows)")
    if lm.dim() != 2 or lm.shape[1] != 1280:
        raise ValueError(f"lm_embeddings rows must have 1280 columns, got {tuple(lm.shape)}")
    return lm


def rec_residue_features(rec: Receptor) -> torch.Tensor:
    """reference datasets/process_mols.py:147-162: [N_res, 1] float32 amino-acid index."""
    return torch.tensor([[safe_index(AMINO_ACIDS, r.resname)] for r in rec.residues], dtype=torch.float32)


def rec_atom_features(resname: str, atom: PdbAtom) -> List[int]:
    """reference datasets/process_mols.py:517-543 (`get_rec_atom_feat`)."""
    element = "C" if atom.element == "CD" else atom.element
    assert element != ""
    atomic_num = ATOMIC_NUMBER.get(element.upper(), -1)
    return [safe_index(AMINO_ACIDS, resname), safe_index(ATOMIC_NUMS, atomic_num),
            safe_index(ATOM_TYPE_2, (atom.name + "*")[:2]), safe_index(ATOM_TYPE_3, atom.name)]


def receptor_graph(rec: Receptor, cutoff: float = 15.0, max_neighbors: Optional[int] = 24, remove_hs: bool = True):
    """reference datasets/process_mols.py:650-723: C-alpha graph (all residues within `cutoff`, the `max_neighbors`
    nearest when there are more, the single nearest when there is none) and the atom -> residue edges.
    Returns (rr_edge_index [2,E], atom_x [N_a,4], atom_pos [N_a,3], atom_res [N_a], heavy mask over all atoms)."""
    ca = rec.ca.astype(np.float64)
    if len(ca) <= 1:
        raise ValueError("rec contains only 1 residue!")
    dist = np.linalg.norm(ca[:, None, :] - ca[None, :, :], axis=-1)
    src, dst = [], []
    for i in range(len(ca)):
        nb = list(np.where(dist[i] < cutoff)[0])
        nb.remove(i)
        if max_neighbors is not None and len(nb) > max_neighbors:
            nb = list(np.argsort(dist[i]))[1:max_neighbors + 1]
        if len(nb) == 0:
            nb = list(np.argsort(dist[i]))[1:2]
        src += [i] * len(nb)
        dst += [int(j) for j in nb]
    feats, pos, res_of = [], [], []
    for ri, r in enumerate(rec.residues):
        for a in r.atoms:
            feats.append(rec_atom_features(r.resname, a))
            pos.append(a.coord)
            res_of.append(ri)
    feats = np.asarray(feats, dtype=np.int64).reshape(-1, 4)
    keep = feats[:, 1] != 0 if remove_hs else np.ones(len(feats), dtype=bool)
    return (np.array([src, dst], dtype=np.int64), feats[keep], np.asarray(pos, dtype=np.float32)[keep],
            np.asarray(res_of, dtype=np.int64)[keep], keep)


# ---------------------------------------------------------------------------------------------- flexible side chains
_GREEK = {"A": "B", "B": "G", "G": "D", "D": "E", "E": "Z", "Z": "H", "H": ""}
_RING_CLOSURES = {("CE1", "NE2"), ("NE1", "CE2"), ("CD2", "CE3"), ("CZ3", "CH2")}


def _keep_sidechain_atom(name: str) -> bool:
    """reference utils/torsion.py:218-222 (`filter_side_chain_atoms`)."""
    return re.search("^(OXT)$|^C$|^O$|^N$|^H|^H$.|^H.$[1-9]", name) is None


def sidechain_rotation_mask(res: PdbResidue, offset: int):
    """reference utils/torsion.py:165-215 + `add_edges` :224-248: directed greek-letter graph over the residue's kept atoms,
    bonds visited in BFS order from CA; a bond is rotatable when removing it splits the graph and the far side has more
    than one atom.  Returns [(atom ids that rotate, [u, v])] with ids = position in the residue + offset."""
    nodes = [a.name for a in res.atoms if _keep_sidechain_atom(a.name)]
    where = [i for i, a in enumerate(res.atoms) if a.name in nodes]
    succ: Dict[str, List[str]] = {v: [] for v in nodes}
    order = list(succ)                      # node order of the graph (duplicates collapse like nx nodes)

    def add(u, v):
        if v not in succ[u]:
            succ[u].append(v)

    for i in range(len(order) - 1):
        for j in range(i + 1, len(order)):
            cur, nxt = order[i], order[j]
            if (cur, nxt) in _RING_CLOSURES:
                add(cur, nxt)
            if len(cur) == len(nxt) == 3:
                if _GREEK[cur[1]] == nxt[1] and cur[2] == nxt[2]:
                    add(cur, nxt)
            elif _GREEK[cur[1]] == nxt[1]:
                add(cur, nxt)
    if "CA" not in succ:
        raise KeyError("CA")
    tree, seen, q = [], {"CA"}, deque(["CA"])           # nx.bfs_tree(G, 'CA').edges()
    while q:
        u = q.popleft()
        for v in succ[u]:
            if v not in seen:
                seen.add(v)
                tree.append((u, v))
                q.append(v)
    idx = {v: k for k, v in enumerate(order)}
    und = [set() for _ in order]
    for u in order:
        for v in succ[u]:
            und[idx[u]].add(idx[v])
            und[idx[v]].add(idx[u])
    out = []
    for u, v in tree:
        und[idx[u]].discard(idx[v])
        und[idx[v]].discard(idx[u])
        comps = _components(len(order), und)
        und[idx[u]].add(idx[v])
        und[idx[v]].add(idx[u])
        if len(comps) < 2:
            continue
        comp = next(c for c in comps if idx[v] in c)
        if len(comp) > 1:
            out.append(([where[k] + offset for k in comp], [where[idx[u]] + offset, where[idx[v]] + offset]))
    return out


def parse_flexible_spec(spec: Optional[str]) -> List[Tuple[str, int]]:
    """'A:160-A:193' (reference inference.py:57) -> [('A', 160), ('A', 193)]."""
    if not spec:
        return []
    return [(p.split(":")[0], int(p.split(":")[1])) for p in spec.split("-") if p]


def sidechain_rotation_masks(rec: Receptor, accept_atom, heavy_mask: Optional[np.ndarray] = None):
    """reference datasets/process_mols.py:773-883: residues with a side-chain atom accepted by `accept_atom(residue, atom)`
    become flexible (ALA / GLY / PRO and non-standard names never); returns (subcomponents, subcomponentsMapping [S,2],
    edge_idx [S,2], residueNBondsMapping, pdb ids), atom ids counted over heavy atoms when `heavy_mask` is given."""
    flex = []
    for r in rec.residues:
        if r.resname in {"ALA", "GLY", "PRO"} or r.resname not in FLEXIBLE_SIDECHAINS:
            continue
        for a in r.atoms:
            if a.element == "H" or a.name in {"CA", "N", "C", "O", "OXT"}:
                continue
            if accept_atom(r, a):
                flex.append((r.chain, r.resseq))
                break
    sub, mapping, edges, n_bonds, ids = [], [], [], [], []
    offset, todo = 0, set(flex)
    for r in rec.residues:
        if (r.chain, r.resseq) in todo:
            todo.discard((r.chain, r.resseq))
            try:
                masks = sidechain_rotation_mask(r, offset)
            except Exception as e:                      # the reference skips such residues with a message (:822-823)
                masks = None
                print(f"Skipping residue {r.resname} {r.chain}:{r.resseq} because of the error:", repr(e))
            if masks is not None:
                n_bonds.append(len(masks))
                ids.append((r.chain, r.resseq))
                for comp, e in masks:
                    mapping.append([len(sub), len(sub) + len(comp)])
                    sub += comp
                    edges.append(e)
        offset += len(r.atoms)
    sub = np.asarray(sub, dtype=np.int64)
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    if heavy_mask is not None:      # remove_hs (:872-881): ids shift down by the hydrogens in front of them
        shift = np.concatenate([[0], np.cumsum(~heavy_mask)])
        sub, edges = sub - shift[sub], edges - shift[edges]
    return sub, np.asarray(mapping, dtype=np.int64).reshape(-1, 2), edges, np.asarray(n_bonds, dtype=np.int64), ids


# ---------------------------------------------------------------------------------------------- the complex graph
def build_complex_graph(pdb_text: str, sdf_text: str, *, name: str = "complex", pocket_center: Optional[Sequence[float]] = None,
                        pocket_reduction: bool = True, pocket_cutoff: float = 5.0, pocket_buffer: float = 10.0,
                        receptor_radius: float = 15.0, c_alpha_max_neighbors: Optional[int] = 24, remove_hs: bool = True,
                        flexible_sidechains: Optional[str] = None, flexdist: Optional[float] = None,
                        lm_embeddings: Optional[np.ndarray] = None, chain_cutoff: float = 10.0) -> HeteroBatch:
    """One complex graph in the schema of SURVEY §8(b), coordinates centred on the pocket centre.

    Defaults are the README model's (reference README.md:72: all atoms, pocket reduction 'center-dist' with buffer 10,
    receptor_radius 15, 24 C-alpha neighbours, hydrogens removed).  `pocket_center` overrides the centre computed from the
    ligand pose (reference pdbbind.py:585-596: the radius is then the ligand's own extent).  Flexible residues: an explicit
    list 'A:160-A:193' (inference.py:57) or every residue with a side-chain atom within `flexdist` of the pocket sphere
    (pdbbind.py:343-349, metric 'L2').  `lm_embeddings`: None (receptor.x then holds the residue index only), or the ESM rows
    in one of three forms: a list with one [chain length, 1280] array per chain of the structure (the reference's
    `lm_embedding_chains`, sliced as process_mols.py:389-397 does: rows of the kept chains, minus the residues the pocket
    selector discarded), ONE array over all chains' amino-acid residues in structure order (sliced the same way), or exactly
    the [n_residues_kept, 1280] rows."""
    mol = parse_sdf(sdf_text)
    x_l, pos_l, ei_l, ea_l, edge_mask, mask_rotate = ligand_graph(mol, keep_hs=not remove_hs)
    residues = parse_pdb(pdb_text)
    lig64 = pos_l.astype(np.float64)

    ca_all = np.array([r.atom("CA").coord for r in residues if r.atom("CA") is not None], dtype=np.float32)
    pocket = None
    if pocket_center is not None:
        centre = np.asarray(pocket_center, dtype=np.float32)
        radius = float(np.linalg.norm(pos_l - pos_l.mean(0, keepdims=True), axis=1).max())
    else:
        centre, radius = binding_pocket(ca_all, pos_l, pocket_cutoff, 0.0)
    radius_buffered = radius + pocket_buffer
    if pocket_reduction:
        pocket = (centre, radius_buffered)
    rec = extract_receptor(residues, lig64, cutoff=chain_cutoff, pocket=pocket)
    rr, atom_x, atom_pos, atom_res, heavy = receptor_graph(rec, receptor_radius, c_alpha_max_neighbors, remove_hs)

    data = HeteroBatch()
    protein_centre = centre if pocket_reduction else rec.ca.mean(0)
    data["ligand"] = Store(x=x_l, pos=torch.from_numpy(pos_l - protein_centre), edge_mask=torch.from_numpy(edge_mask),
                           mask_rotate=mask_rotate)
    data["ligand", "ligand"] = Store(edge_index=torch.from_numpy(ei_l), edge_attr=torch.from_numpy(ea_l))
    res_x = rec_residue_features(rec)
    if lm_embeddings is not None:
        res_x = torch.cat([res_x, slice_lm_embeddings(lm_embeddings, rec)], 1)
    data["receptor"] = Store(x=res_x, pos=torch.from_numpy(rec.ca - protein_centre))
    data["receptor", "receptor"] = Store(edge_index=torch.from_numpy(rr))
    data["atom"] = Store(x=torch.from_numpy(atom_x), pos=torch.from_numpy(atom_pos - protein_centre))
    data["atom", "receptor"] = Store(edge_index=torch.from_numpy(np.stack([np.arange(len(atom_res)), atom_res])))

    wanted = set(parse_flexible_spec(flexible_sidechains))
    accept = None
    if wanted:
        accept = lambda r, a: (r.chain, r.resseq) in wanted                                  # noqa: E731
    elif flexdist is not None:
        lim = radius + flexdist
        accept = lambda r, a: float(np.linalg.norm(a.coord - centre)) < lim                   # noqa: E731
    if accept is not None:
        sub, mapping, edges, n_bonds, ids = sidechain_rotation_masks(rec, accept, heavy if remove_hs else None)
        if edges.shape[0] > 0:
            st = Store(edge_idx=torch.from_numpy(edges), subcomponents=torch.from_numpy(sub),
                       subcomponentsMapping=torch.from_numpy(mapping), residueNBondsMapping=torch.from_numpy(n_bonds), pdbIds=ids)
            st.num_nodes = edges.shape[0]
            data["flexResidues"] = st
    data.num_graphs = 1
    data.name = name
    data.original_center = torch.from_numpy(np.asarray(protein_centre, dtype=np.float32)).reshape(1, 3)
    return data
