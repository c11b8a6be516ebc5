"""Build the 3dpf benchmark-complex geometry fixture from the reference's example data.

TEST INFRASTRUCTURE (oracle/): run ONCE in the build container (needs /root/reference and
networkx); the product and the GPU box only ever read the resulting small .npz (data, no code).

What it reproduces (plain text parsing, no rdkit/biopython):
  * ligand heavy atoms + directed bond list in the order of reference
    datasets/process_mols.py:435-453 (row += [start,end]; col += [end,start]),
    bond one-hot of 4 (`bonds` dict, process_mols.py:67);
  * rotatable-bond `edge_mask` / `mask_rotate` following reference utils/torsion.py:16-65;
  * pocket reduction of reference datasets/pdbbind.py:324-339,585-603,775-784
    (centre = mean CA within 5 A of any ligand atom, radius = max|lig-c| + 10 A, a residue is kept if
    any of its atoms is inside), then centring on the pocket centre (pdbbind.py:704-731);
  * receptor CA graph of reference datasets/process_mols.py:650-700 with the README training
    settings (receptor_radius 15, c_alpha_max_neighbors 24, README.md:72);
  * atom -> residue edges (process_mols.py:715-723);
  * flexible side chains A:160,193,197,198,222,224,227 (README.md:47): chi-bond list, rotated
    sub-components and their mapping in the layout of process_mols.py:888-912 /
    utils/torsion.py:165-215 (greek-letter side-chain graph, BFS from CA).
Categorical node features that would need rdkit/biopython featurisers are NOT produced here; the
synthetic generator draws them at random within the reference's feature dims.

Usage: python oracle/make_3dpf_geometry.py [out.npz]
"""
import os
import re
import sys

import numpy as np

REF = os.environ.get("DDP_REFERENCE", "/root/reference")
FLEX = [160, 193, 197, 198, 222, 224, 227]
BOND_TYPE = {1: 0, 2: 1, 3: 2, 4: 3}  # SDF bond order -> index into reference `bonds` dict


def parse_sdf(path):
    lines = open(path).read().splitlines()
    na, nb = int(lines[3][0:3]), int(lines[3][3:6])
    pos, elem = [], []
    for ln in lines[4:4 + na]:
        pos.append([float(ln[0:10]), float(ln[10:20]), float(ln[20:30])])
        elem.append(ln[31:34].strip())
    bonds = []
    for ln in lines[4 + na:4 + na + nb]:
        bonds.append((int(ln[0:3]) - 1, int(ln[3:6]) - 1, int(ln[6:9])))
    return np.array(pos, np.float64), elem, bonds


def parse_pdb(path):
    atoms = []
    for ln in open(path):
        if not ln.startswith("ATOM"):
            continue
        name = ln[12:16].strip()
        resn = ln[17:20].strip()
        chain = ln[21]
        resi = int(ln[22:26])
        xyz = [float(ln[30:38]), float(ln[38:46]), float(ln[46:54])]
        el = ln[76:78].strip() or name[0]
        atoms.append((chain, resi, resn, name, el, xyz))
    return atoms


def transformation_mask(n_atoms, dir_edges):
    """reference utils/torsion.py:16-65 on the directed bond list (pairs of consecutive entries)."""
    import networkx as nx
    G = nx.Graph()
    G.add_nodes_from(range(n_atoms))
    G.add_edges_from([tuple(e) for e in dir_edges])
    to_rotate = []
    for i in range(0, len(dir_edges), 2):
        assert dir_edges[i][0] == dir_edges[i + 1][1]
        G2 = G.copy()
        G2.remove_edge(*dir_edges[i])
        if not nx.is_connected(G2):
            comp = list(sorted(nx.connected_components(G2), key=len)[0])
            if len(comp) > 1:
                if dir_edges[i][0] in comp:
                    to_rotate.append([])
                    to_rotate.append(comp)
                else:
                    to_rotate.append(comp)
                    to_rotate.append([])
                continue
        to_rotate.append([])
        to_rotate.append([])
    mask_edges = np.asarray([len(l) > 0 for l in to_rotate], dtype=bool)
    mask_rotate = np.zeros((int(mask_edges.sum()), n_atoms), dtype=bool)
    idx = 0
    for i in range(len(dir_edges)):
        if mask_edges[i]:
            mask_rotate[idx][np.asarray(to_rotate[i], dtype=int)] = True
            idx += 1
    return mask_edges, mask_rotate


_ORDER = {"A": "B", "B": "G", "G": "D", "D": "E", "E": "Z", "Z": "H", "H": ""}


def _keep_sc(name):  # reference utils/torsion.py:218-222
    return re.search("^(OXT)$|^C$|^O$|^N$|^H|^H$.|^H.$[1-9]", name) is None


def sidechain_masks(res_atom_names, offset):
    """reference utils/torsion.py:165-248 for one residue; returns [(rotated atom ids, [u, v])]."""
    import networkx as nx
    nodes = [n for n in res_atom_names if _keep_sc(n)]
    heavy = [i for i, n in enumerate(res_atom_names) if n in nodes]
    G = nx.DiGraph()
    G.add_nodes_from(nodes)
    for i in range(len(nodes) - 1):
        for j in range(i + 1, len(nodes)):
            for cur, nxt in ((nodes[i], nodes[j]), (nodes[j], nodes[i])):
                if (cur, nxt) in (("CE1", "NE2"), ("NE1", "CE2"), ("CD2", "CE3"), ("CZ3", "CH2")):
                    G.add_edge(cur, nxt)
                elif len(cur) == len(nxt) == 3:
                    if len(cur) > 1 and _ORDER.get(cur[1], None) == nxt[1] and cur[2] == nxt[2]:
                        G.add_edge(cur, nxt)
                elif len(cur) > 1 and len(nxt) > 1 and _ORDER.get(cur[1], None) == nxt[1]:
                    G.add_edge(cur, nxt)
    out = []
    if "CA" not in G:
        return out
    for edge in nx.bfs_tree(G, "CA").edges():
        G2 = G.to_undirected()
        G2.remove_edge(*edge)
        if nx.is_connected(G2):
            continue
        comps = list(nx.connected_components(G2))
        comp = next(c for c in comps if edge[1] in c)
        if len(comp) > 1:
            g2 = list(G2.nodes)
            rot = [heavy[g2.index(v)] + offset for v in comp]
            out.append((sorted(rot), [heavy[g2.index(edge[0])] + offset, heavy[g2.index(edge[1])] + offset]))
    return out


def main():
    out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else
                          os.path.join(os.path.dirname(__file__), "..", "diffdock_pocket_amd", "assets", "3dpf_geometry.npz"))
    lpos, lel, lbonds = parse_sdf(os.path.join(REF, "example_data", "3dpf_ligand.sdf"))
    heavy = [i for i, e in enumerate(lel) if e != "H"]
    remap = {a: k for k, a in enumerate(heavy)}
    lig_pos = lpos[heavy]
    row, col, btype = [], [], []
    for a, b, o in lbonds:
        if a in remap and b in remap:
            row += [remap[a], remap[b]]
            col += [remap[b], remap[a]]
            btype += [BOND_TYPE.get(o, 0)] * 2
    lig_edge_index = np.array([row, col], np.int64)
    lig_edge_attr = np.eye(4, dtype=np.float32)[np.array(btype)]
    edge_mask, mask_rotate = transformation_mask(len(heavy), list(zip(row, col)))

    atoms = [a for a in parse_pdb(os.path.join(REF, "example_data", "3dpf_protein.pdb")) if a[4] != "H"]
    residues = {}
    for a in atoms:
        residues.setdefault((a[0], a[1], a[2]), []).append(a)
    res_keys = [k for k in residues if any(a[3] == "CA" for a in residues[k])]
    ca_all = np.array([[a[5] for a in residues[k] if a[3] == "CA"][0] for k in res_keys])
    d = np.linalg.norm(ca_all[:, None] - lig_pos[None], axis=-1)
    label = (d < 5.0).any(1)
    centre = ca_all[label].mean(0)
    radius = np.linalg.norm(lig_pos - centre[None], axis=1).max() + 10.0
    keep = [k for k in res_keys
            if (np.linalg.norm(np.array([a[5] for a in residues[k]]) - centre, axis=1) < radius).any()]

    rec_pos = np.array([[a[5] for a in residues[k] if a[3] == "CA"][0] for k in keep]) - centre
    atom_pos, atom_res, atom_names, atom_elem, res_names = [], [], [], [], []
    flex_sub, flex_map, flex_edges, off = [], [], [], 0
    for ri, k in enumerate(keep):
        names = [a[3] for a in residues[k]]
        if k[1] in FLEX:
            for comp, e in sidechain_masks(names, off):
                flex_map.append([len(flex_sub), len(flex_sub) + len(comp)])
                flex_sub += comp
                flex_edges.append(e)
        for a in residues[k]:
            atom_pos.append(a[5]); atom_res.append(ri); atom_names.append(a[3]); atom_elem.append(a[4])
        res_names.append(k[2])
        off += len(names)
    atom_pos = np.array(atom_pos) - centre

    # receptor CA graph: <=24 nearest within 15 A (process_mols.py:661-681)
    D = np.linalg.norm(rec_pos[:, None] - rec_pos[None], axis=-1)
    src, dst = [], []
    for i in range(len(rec_pos)):
        nb = list(np.where(D[i] < 15.0)[0]); nb.remove(i)
        if len(nb) > 24:
            nb = list(np.argsort(D[i]))[1:25]
        if len(nb) == 0:
            nb = list(np.argsort(D[i]))[1:2]
        src += [i] * len(nb); dst += nb

    np.savez_compressed(
        out,
        lig_pos=(lig_pos - centre).astype(np.float32), lig_elem=np.array([lel[i] for i in heavy]),
        lig_edge_index=lig_edge_index, lig_edge_attr=lig_edge_attr, lig_edge_mask=edge_mask,
        lig_mask_rotate=mask_rotate,
        rec_pos=rec_pos.astype(np.float32), rec_resname=np.array(res_names),
        rec_edge_index=np.array([src, dst], np.int64),
        atom_pos=atom_pos.astype(np.float32), atom_res=np.array(atom_res, np.int64),
        atom_name=np.array(atom_names), atom_elem=np.array(atom_elem),
        flex_edge_idx=np.array(flex_edges, np.int64).reshape(-1, 2),
        flex_subcomponents=np.array(flex_sub, np.int64),
        flex_subcomponents_mapping=np.array(flex_map, np.int64).reshape(-1, 2),
        pocket_centre=centre.astype(np.float32), pocket_radius=np.float32(radius),
    )
    print(f"wrote {out}: N_l={len(heavy)} E_bond={lig_edge_index.shape[1]} T={int(edge_mask.sum())} "
          f"N_r={len(rec_pos)} N_a={len(atom_pos)} E_rr={len(src)} S={len(flex_edges)}")


if __name__ == "__main__":
    main()
