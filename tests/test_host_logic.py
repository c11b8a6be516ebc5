"""CPU tests of the host side: weight packing / shape descriptors (through a pure-torch emulation of the kernel's
documented data flow), graph construction against the oracle's restated torch_cluster semantics, state_dict layout,
C-ABI exports, and the no-fallback rule."""
import ctypes
import functools
import os
import re

import pytest
import torch

from diffdock_pocket_amd import _lib as L
from diffdock_pocket_amd import graph as G
from diffdock_pocket_amd import packing as P
from diffdock_pocket_amd.batch import collate
from diffdock_pocket_amd.score_model import TensorProductScoreModel
from oracle import thirdparty as tp
from oracle.cases import CASES
from oracle.ref_model import OracleConfig, faster_tensor_product

from helpers import load_golden, golden_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def unpack_tiles(flat, ntiles, kp):
    """inverse of the layout documented in csrc/ddp_conv.hip: -> [ntiles*32 columns, kp]"""
    W = flat.reshape(ntiles, kp // 8, 2, 32, 4)            # [tile, m, hh, j, i]
    return W.permute(0, 3, 1, 2, 4).reshape(ntiles * 32, kp)


def emulate_conv(spec, w1p, b1p, w2p, b2p, edge_attr, x_src_rows, sh, fact=None):
    """What ddp_conv_messages_kernel computes for one conv, written with dense torch ops on the PACKED weights."""
    E = edge_attr.shape[0]
    W1 = unpack_tiles(w1p, spec.nct1, spec.kp1)            # [cols, kp1]
    ea = torch.zeros(E, spec.kp1, dtype=torch.float64)
    ea[:, :spec.f_in] = edge_attr
    h = torch.relu(ea @ W1.double().T + b1p.double())[:, :spec.hp]
    hp = torch.zeros(E, spec.hp, dtype=torch.float64)
    hp[:, :h.shape[1]] = h
    hp[:, spec.hid:] = 0
    W2 = unpack_tiles(w2p, spec.ntiles, spec.hp).double()
    acc = hp @ W2.T + b2p.double()                          # [E, ntiles*32]
    out = torch.zeros(E, spec.d_out, dtype=torch.float64)
    s0, s1 = sh[:, 0].double(), sh[:, 1:4].double()
    x = x_src_rows.double()
    if fact is not None:   # source-node factorised part: tv[e] = h[e] . G[src(e)] + Gb[src(e)], times s0 / s1[c]
        Wg, Bg, offs = fact
        for b in spec.blocks:
            if b.g_slot < 0:
                continue
            xs = x[:, offs[b.g_slot]:offs[b.g_slot] + Wg[b.g_slot].shape[0]]
            hg = (spec.hid + 3) // 4 * 4      # G[j][k/4][c][k%4] -> [E, hid, g_cols]
            gc_ = spec.g_cols[b.g_slot]
            row = xs @ Wg[b.g_slot].double()                       # [G | Gb | padding] per node
            G = row[:, :hg * gc_].reshape(E, hg // 4, gc_, 4).permute(0, 1, 3, 2)
            G = G.reshape(E, hg, gc_)[:, :spec.hid]
            Gb = row[:, hg * gc_:(hg + 1) * gc_]
            assert torch.equal(Gb, xs @ Bg[b.g_slot].double()) and float(row[:, (hg + 1) * gc_:].abs().sum()) == 0
            tv = torch.einsum("ek,ekn->en", hp[:, :spec.hid], G) + Gb
            tv = tv[:, b.g_col0:b.g_col0 + b.n]
            for c in range(b.C):
                fac = s0 if b.C == 1 else s1[:, c]
                out[:, b.out_off + torch.arange(b.n) * b.C + c] += fac[:, None] * tv
    for b in spec.blocks:
        if b.U == 0:
            continue
        feats = []
        for kind, off, cnt in b.segs:
            if kind == L.F_SCALAR_S0:
                feats.append((x[:, off:off + cnt] * s0[:, None])[:, :, None])
            elif kind == L.F_DOT:
                a = x[:, off:off + 3 * cnt].reshape(E, cnt, 3)
                feats.append(((a * s1[:, None]).sum(-1) / 3 ** 0.5)[:, :, None])
            elif kind == L.F_SCALAR_S1:
                feats.append(x[:, off:off + cnt, None] * s1[:, None, :])
            elif kind == L.F_VEC_S0:
                feats.append(x[:, off:off + 3 * cnt].reshape(E, cnt, 3) * s0[:, None, None])
            else:
                a = x[:, off:off + 3 * cnt].reshape(E, cnt, 3)
                feats.append(torch.linalg.cross(a, s1[:, None].expand_as(a), dim=-1) / 2 ** 0.5)
        F = torch.cat(feats, 1)                             # [E, U, C]
        assert F.shape[1] == b.U and F.shape[2] == b.C
        for t in range(b.ntiles):
            for j in range(32):
                if b.nsub > 1:
                    u, sub = divmod(t, b.nsub)
                    ncol = sub * 32 + j
                    valid = ncol < b.n and u < b.U
                else:
                    us, ncol = divmod(j, b.n)
                    u = t * b.ups + us
                    valid = us < b.ups and u < b.U
                if not valid:
                    assert float(acc[:, (b.tile0 + t) * 32 + j].abs().max()) == 0.0  # padded columns are exactly zero
                    continue
                for c in range(b.C):
                    out[:, b.out_off + ncol * b.C + c] += F[:, u, c] * acc[:, (b.tile0 + t) * 32 + j]
    return out


@pytest.mark.parametrize("factorized", [False, True])
@pytest.mark.parametrize("ns,nv,layer", [(16, 4, 0), (16, 4, 1), (24, 6, 2), (60, 10, 3), (60, 10, 0), (32, 6, 3)])
def test_packed_conv_matches_faster_tensor_product(ns, nv, layer, factorized):
    torch.manual_seed(layer + ns)
    mi, mo = P.irreps_muls(ns, nv, layer), P.irreps_muls(ns, nv, layer + 1)
    spec = P.faster_tp_spec(mi, mo, 3 * ns, factorized=factorized)
    cfg = OracleConfig(ns=ns, nv=nv)
    E = 7
    fc0_w, fc0_b = torch.randn(3 * ns, 3 * ns) / (3 * ns) ** 0.5, torch.randn(3 * ns) * 0.1
    fc3_w, fc3_b = torch.randn(spec.weight_numel, 3 * ns) / (3 * ns) ** 0.5, torch.randn(spec.weight_numel) * 0.1
    ea, x, sh = torch.randn(E, 3 * ns), torch.randn(E, P.irreps_dim(mi)), torch.randn(E, 4)
    w1p, b1p = P.pack_fc1(spec, fc0_w, fc0_b)
    w2p, b2p = P.pack_fc2(spec, fc3_w, fc3_b)
    fact = P.factor_weights(spec, fc3_w, fc3_b) if factorized else None
    got = emulate_conv(spec, w1p, b1p, w2p, b2p, ea, x, sh, fact)
    w = torch.relu(ea.double() @ fc0_w.double().T + fc0_b.double()) @ fc3_w.double().T + fc3_b.double()
    want = faster_tensor_product(cfg.irreps(layer), cfg.irreps(layer + 1), x.double(), sh.double(), w)
    assert torch.allclose(got, want, rtol=2e-5, atol=2e-6)
    assert spec.weight_numel == fc3_w.shape[0]


def test_final_conv_and_torsion_specs():
    ns, nv = 16, 4
    m = P.irreps_muls(ns, nv, 3)
    spec = P.faster_tp_spec(m, (0, 2, 2, 0), 2 * ns)
    assert spec.d_out == 12 and [b.ups for b in spec.blocks] == [16, 16]
    E = 5
    torch.manual_seed(0)
    fc0_w, fc0_b = torch.randn(2 * ns, 2 * ns), torch.randn(2 * ns)
    fc3_w, fc3_b = torch.randn(spec.weight_numel, 2 * ns), torch.randn(spec.weight_numel)
    ea, x, sh = torch.randn(E, 2 * ns), torch.randn(E, P.irreps_dim(m)), torch.randn(E, 4)
    got = emulate_conv(spec, *P.pack_fc1(spec, fc0_w, fc0_b), *P.pack_fc2(spec, fc3_w, fc3_b), ea, x, sh)
    w = torch.relu(ea.double() @ fc0_w.double().T + fc0_b.double()) @ fc3_w.double().T + fc3_b.double()
    want = faster_tensor_product(OracleConfig(ns=ns, nv=nv).irreps(3), "2x1o+2x1e", x.double(), sh.double(), w)
    assert torch.allclose(got, want, rtol=2e-5, atol=2e-6)
    # torsion conv against the restated e3nn FullyConnectedTensorProduct
    tspec = P.torsion_tp_spec(m, ns, 3 * ns)
    fctp = tp.FullyConnectedTensorProduct(OracleConfig(ns=ns, nv=nv).irreps(3), "1x1o+1x2e+1x2o+1x3o", f"{ns}x0o+{ns}x0e")
    assert tspec.weight_numel == fctp.weight_numel
    fc0_w, fc0_b = torch.randn(3 * ns, 3 * ns), torch.randn(3 * ns)
    fc3_w, fc3_b = torch.randn(tspec.weight_numel, 3 * ns), torch.randn(tspec.weight_numel)
    ea = torch.randn(E, 3 * ns)
    t = torch.randn(E, 3)
    tor_sh20 = torch.cat([t, torch.zeros(E, 17)], 1)
    got = emulate_conv(tspec, *P.pack_fc1(tspec, fc0_w, fc0_b), *P.pack_fc2(tspec, fc3_w, fc3_b), ea, x,
                       torch.cat([torch.zeros(E, 1), t], 1))
    w = torch.relu(ea.double() @ fc0_w.double().T + fc0_b.double()) @ fc3_w.double().T + fc3_b.double()
    want = fctp(x.double(), tor_sh20.double(), w)
    assert torch.allclose(got, want, rtol=2e-5, atol=2e-6)


def test_bn_affine_matches_restated_batchnorm():
    torch.manual_seed(1)
    ns, nv = 8, 3
    blocks = [(ns, 1, True), (nv, 3, False), (nv, 3, False), (ns, 1, False)]
    nf, nsc = 2 * ns + 2 * nv, ns
    rm, rv, w, b = torch.randn(nsc), torch.rand(nf) + 0.5, torch.rand(nf) + 0.5, torch.randn(nsc)
    x = torch.randn(11, 2 * ns + 6 * nv)
    sc, sh = P.bn_affine(blocks, rm, rv, w, b)
    want = tp.batch_norm_eval(f"{ns}x0e+{nv}x1o+{nv}x1e+{ns}x0o", x, rm, rv, w, b)
    assert torch.allclose(x * sc + sh, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("uniform", [True, False])
def test_graph_builders_match_oracle_semantics(uniform):
    torch.manual_seed(2)
    sizes_x = [40, 40, 40] if uniform else [40, 17, 33]
    sizes_y = [9, 9, 9] if uniform else [9, 4, 12]
    x = torch.cat([torch.randn(n, 3) * 4 for n in sizes_x])
    y = torch.cat([torch.randn(n, 3) * 4 for n in sizes_y])
    bx = torch.repeat_interleave(torch.arange(3), torch.tensor(sizes_x))
    by = torch.repeat_interleave(torch.arange(3), torch.tensor(sizes_y))
    lx, ly = G.DenseLayout.build(bx, 3), G.DenseLayout.build(by, 3)
    for cap in (10000, 5):
        for rule in (None, "first_index", "nearest"):
            a = G.radius(x, y, 3.0, lx, ly, max_num_neighbors=cap, truncation=rule)
            b = tp.radius(x, y, 3.0, bx, by, max_num_neighbors=cap, truncation=rule)
            assert torch.equal(a, b)
    assert torch.equal(G.radius_graph(x, 3.0, lx), tp.radius_graph(x, 3.0, bx))
    for rule in ("first_index", "nearest"):
        assert torch.equal(G.radius_graph(x, 3.0, lx, max_num_neighbors=4, truncation=rule),
                           tp.radius_graph(x, 3.0, bx, max_num_neighbors=4, truncation=rule))
    a, b = G.knn_graph(x, 8, lx), tp.knn_graph(x, 8, bx)
    assert torch.equal(a, b)
    csr = G.build_csr(a[0], a[1], x.shape[0])
    assert int(csr.rowptr[-1]) == a.shape[1]
    assert torch.equal(csr.recv.long(), a[0][csr.eid.long()]) and torch.equal(csr.src.long(), a[1][csr.eid.long()])
    assert bool((csr.recv[1:] >= csr.recv[:-1]).all())


@pytest.mark.parametrize("name", list(CASES))
def test_state_dict_layout_matches_reference(name):
    """Keys and shapes equal those of the reference's own module tree (captured in the golden file): reference
    checkpoints load with strict=True."""
    case, gold = CASES[name], load_golden(name)
    kw = dict(case.model_kwargs())
    kw.update(case.ctor_extras())
    model = TensorProductScoreModel(**kw)
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert mine == gold["state_dict_shapes"]
    sd = golden_state_dict(gold, case.weight_seed)
    sd["final_tp_tor.some_e3nn_buffer"] = torch.zeros(3)       # e3nn-internal buffers of real checkpoints are ignored
    model.load_state_dict(sd, strict=True)
    for k, v in gold["offsets"].items():
        assert torch.equal(model.state_dict()[k], v)


def test_unsupported_configs_raise():
    case = CASES["cfg1_full"]
    for bad in ({"sh_lmax": 2}, {"use_second_order_repr": True}, {"odd_parity": True}, {"affinity_prediction": True, "confidence_mode": True}):
        kw = dict(case.model_kwargs())
        kw.update(case.ctor_extras())
        kw.update(bad)
        with pytest.raises(NotImplementedError):
            TensorProductScoreModel(**kw)


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU instead of computing somewhere else."""
    case = CASES["cfg1_edge"]
    kw = dict(case.model_kwargs())
    kw.update(case.ctor_extras())
    model = TensorProductScoreModel(**kw).eval()
    with pytest.raises(L.DdpError):
        model(case.make_batch())
    src = open(os.path.join(ROOT, "diffdock_pocket_amd", "score_model.py")).read()
    assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# ", "")
    for fn in os.listdir(os.path.join(ROOT, "diffdock_pocket_amd")):
        if fn.endswith(".py"):
            body = open(os.path.join(ROOT, "diffdock_pocket_amd", fn)).read()
            assert "import oracle" not in body and "from oracle" not in body, fn


def test_c_abi_exports_every_declared_symbol():
    from diffdock_pocket_amd import build
    build.build(verbose=False)
    header = open(os.path.join(ROOT, "include", "ddp_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(ddp_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared == set(L.EXPORTS)
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.ddp_abi_version.restype = ctypes.c_int
    assert lib.ddp_abi_version() == L.load().ddp_abi_version() == 17
    assert ctypes.sizeof(L.ConvShape) == 11 * 4 + 4 * (11 * 4 + 3 * 12) + 4 + 4 * 4 + 4 * 2 * 20


def test_shared_receptor_side_detection():
    """score_model._shared_receptor_side (pure host logic): N copies of one complex -> every receptor-side conv is marked
    shared with the per-graph node / edge counts; any difference between the graphs un-marks exactly the convs it touches."""
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    case = CASES["cfg1_edge"]
    kw = dict(case.model_kwargs())
    kw.update(case.ctor_extras())
    model = TensorProductScoreModel(**kw)
    B = 3
    gs = [make_3dpf_complex(seed=1, flexible_sidechains=True, n_rec=12) for _ in range(B)]
    b = collate(gs)
    rec, atom = b["receptor"], b["atom"]
    nr, na = rec.pos.shape[0] // B, atom.pos.shape[0] // B
    lay_r, lay_a = G.DenseLayout.build(rec.batch.long(), B), G.DenseLayout.build(atom.batch.long(), B)
    rr, ar = b["receptor", "receptor"].edge_index.long(), b["atom", "receptor"].edge_index.long()

    def detect():
        aa = G.knn_graph(atom.pos.float(), 8, lay_a)
        return model._shared_receptor_side(B, rec, atom, rec.pos.float(), atom.pos.float(), lay_r, lay_a, rr, ar, aa), aa

    out, aa = detect()
    assert out[6] == (nr, rr.shape[1] // B, nr)
    assert out[3] == (na, aa.shape[1] // B, na)
    assert out[5] == (na, ar.shape[1] // B, nr) and out[8] == (nr, ar.shape[1] // B, na)
    atom.pos[na + 5] += 0.25                         # one atom of graph 1 moves: atom-side convs lose the shortcut
    out, _ = detect()
    assert out[6] is not None and out[3] is None and out[5] is None and out[8] is None
    atom.pos[na + 5] -= 0.25
    rec.x[2 * nr + 1, 0] += 1                        # one residue type of graph 2 differs
    out, _ = detect()
    assert out[6] is None and out[5] is None and out[8] is None and out[3] is not None
    rec.x[2 * nr + 1, 0] -= 1
    rr2 = rr.clone()
    rr2[1, -1] = rr2[1, -2]                          # one edge of the last graph differs
    out = model._shared_receptor_side(B, rec, atom, rec.pos.float(), atom.pos.float(), lay_r, lay_a, rr2, ar, aa)
    assert out[6] is None and out[3] is not None and out[5] is not None
    assert model._shared_receptor_side(1, rec, atom, rec.pos.float(), atom.pos.float(), lay_r, lay_a, rr, ar, aa) \
        == {3: None, 5: None, 6: None, 8: None}
    # atoms="static" (flexible side chains): positions are not examined, features and atom-receptor edges are
    atom.pos[na + 5] += 0.25
    out = model._shared_receptor_side(B, rec, atom, rec.pos.float(), atom.pos.float(), lay_r, lay_a, rr, ar, aa, atoms="static")
    assert out[6] is not None and out[3] is None and out["flex"] == (na, ar.shape[1] // B, nr)
    atom.x[na + 5, 0] += 1
    out = model._shared_receptor_side(B, rec, atom, rec.pos.float(), atom.pos.float(), lay_r, lay_a, rr, ar, aa, atoms="static")
    assert out[6] is not None and "flex" not in out


def test_split_bf16x3_and_sigma_range_detection():
    """packing.split_bf16x3: the three bfloat16 planes add up to the fp32 weights (to 2^-24 relative) in the operand layout of
    include/ddp_hip.h; score_model._sigma_ranges: only the package's own schedule bound to its ranges is evaluated in the kernel."""
    from diffdock_pocket_amd.diffusion import SigmaRanges, t_to_sigma
    torch.manual_seed(0)
    W = torch.randn(2, 60, 96) * torch.exp(torch.randn(2, 1, 96))
    W3 = P.split_bf16x3(W)
    assert W3.dtype == torch.bfloat16 and W3.shape == (2, 3, 4, 2, 96, 8)
    back = W3.float().permute(0, 1, 2, 3, 5, 4).reshape(2, 3, 64, 96).sum(1)
    assert torch.equal(back[:, 60:], torch.zeros(2, 4, 96))
    assert float(((back[:, :60] - W).abs() / W.abs().clamp_min(1e-30)).max()) < 2.0 ** -22
    case = CASES["cfg1_edge"]
    kw = dict(case.model_kwargs())
    kw.update(case.ctor_extras())
    model = TensorProductScoreModel(**kw)
    rng = SigmaRanges()
    model.t_to_sigma = functools.partial(t_to_sigma, args=rng)
    assert model._sigma_ranges() == ((rng.tr_sigma_min, rng.tr_sigma_max), (rng.rot_sigma_min, rng.rot_sigma_max),
                                     (rng.tor_sigma_min, rng.tor_sigma_max), (rng.sidechain_tor_sigma_min, rng.sidechain_tor_sigma_max))
    model.t_to_sigma = lambda *ts: tuple(t * 2 for t in ts)
    assert model._sigma_ranges() is None
    model.t_to_sigma = functools.partial(t_to_sigma, args=object())
    assert model._sigma_ranges() is None


@pytest.mark.parametrize("key", ["score_README_72", "confidence_README_88", "confidence_two_cutoffs", "score_old_yml"])
def test_get_model_passes_the_reference_kwargs(key, monkeypatch):
    """factory.get_model against the kwargs the reference's OWN get_model (utils/utils.py:59-113) passes for namespaces made
    by the reference's own parsers from the README command lines (tests/golden/factory_kwargs.json, captured by
    oracle/make_golden_factory.py): every kwarg equal, incl. num_confidence_outputs = len(cutoffs) + 1 and the `in`-guard
    defaults of old yml files; the time embedding is compared through its values on a probe."""
    import argparse
    import json
    from diffdock_pocket_amd import factory
    with open(os.path.join(ROOT, "tests", "golden", "factory_kwargs.json")) as f:
        gold = json.load(f)[key]
    seen = {}

    class Rec:
        def __init__(self, **kw):
            seen.update(kw)

        def to(self, device):
            return self

    monkeypatch.setattr(factory, "TensorProductScoreModel", Rec)
    args = argparse.Namespace(**gold["args"])
    conf = gold["kwargs"]["confidence_mode"]
    factory.get_model(args, torch.device("cpu"), "T2S", no_parallel=True, confidence_mode=conf)
    assert seen.pop("t_to_sigma") == "T2S" and seen.pop("device") == torch.device("cpu")
    emb = seen.pop("timestep_emb_func")
    probe = gold["timestep_emb_probe"]
    assert torch.allclose(emb(torch.tensor(probe["t"])), torch.tensor(probe["values"]), atol=1e-6)
    assert set(seen) == set(gold["kwargs"]), set(seen) ^ set(gold["kwargs"])
    for k, v in gold["kwargs"].items():
        assert seen[k] == v, (k, seen[k], v)


def test_get_model_guards():
    import argparse
    import json
    from diffdock_pocket_amd import factory
    with open(os.path.join(ROOT, "tests", "golden", "factory_kwargs.json")) as f:
        a = json.load(f)["score_README_72"]["args"]
    with pytest.raises(NotImplementedError):       # coarse-grained model: out of scope, refused loudly
        factory.get_model(argparse.Namespace(**dict(a, all_atoms=False)), torch.device("cpu"), None, no_parallel=True)
    with pytest.raises(NotImplementedError):       # the reference would wrap in PyG DataParallel here (utils/utils.py:110)
        factory.get_model(argparse.Namespace(**a), torch.device("cuda:0"), None, no_parallel=False)
    m = factory.get_model(argparse.Namespace(**a), torch.device("cpu"), None, no_parallel=True)
    assert isinstance(m, TensorProductScoreModel) and m.ns == 60 and m.num_conv_layers == 6 and m.flexible_sidechains


def test_library_carries_the_hash_of_the_sources_in_the_tree():
    """build.py compiles the SHA-256 of the kernel sources into the library; _lib.load() refuses a library built from other
    sources, so the binary the GPU tests run is provably the one of the checked-out tree."""
    from diffdock_pocket_amd import build as B
    lib = L.load()
    assert lib.ddp_source_hash().decode() == B.source_hash() and len(B.source_hash()) == 16
    assert not B.needs_build()


def test_radius_cap_keeps_the_first_matches_by_index_like_torch_cluster_on_a_gpu():
    """The side-chain torsion head searches the pocket atoms around a bond centre with the default cap of 32, and bonds do
    see that many atoms within 5 A (3dpf's flexible bonds: 10 .. 32, SURVEY Appendix B.3: up to ~40 on other complexes).
    The reference runs torch_cluster's CUDA kernel there, which scans the atoms in index order and stops at the cap: the
    default rule here.  Pinned on the 3dpf pocket with the cap lowered to 16 so that it binds: a bond centre with more
    candidates gets exactly the 16 lowest-index ones, in index order; the 'nearest' rule gives another set."""
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    g = make_3dpf_complex(seed=0, flexible_sidechains=True)
    pos = g["atom"].pos
    fr = g["flexResidues"]
    mid = (pos[fr.edge_idx[:, 0]] + pos[fr.edge_idx[:, 1]]) / 2
    la = G.DenseLayout.build(torch.zeros(pos.shape[0], dtype=torch.long), 1)
    lb = G.DenseLayout.build(torch.zeros(mid.shape[0], dtype=torch.long), 1)
    assert G.TRUNCATION == "first_index" and tp.TRUNCATION == "first_index"
    full = G.radius(pos, mid, 5.0, la, lb, max_num_neighbors=10000)
    n_within = torch.bincount(full[0], minlength=mid.shape[0])
    cap = 16
    assert int(n_within.max()) >= 32 and int((n_within > cap).sum()) >= 10
    capped = G.radius(pos, mid, 5.0, la, lb, max_num_neighbors=cap)                       # default rule
    assert torch.equal(capped, tp.radius(pos, mid, 5.0, max_num_neighbors=cap))
    near = G.radius(pos, mid, 5.0, la, lb, max_num_neighbors=cap, truncation="nearest")
    assert torch.equal(near, tp.radius(pos, mid, 5.0, max_num_neighbors=cap, truncation="nearest"))
    differs = 0
    for q in range(mid.shape[0]):
        want = full[1][full[0] == q][:cap]                          # ascending index, first `cap`
        assert torch.equal(capped[1][capped[0] == q], want)
        differs += int(not torch.equal(near[1][near[0] == q], want))
    assert differs > 0


def test_set_time_builds_one_tensor_per_distinct_value():
    """set_time (reference utils/diffusion_utils.py:124-165): constant per-node / per-graph time tensors.  Equal python values
    share one tensor per store (nothing downstream writes to them); distinct values and tensor inputs get their own."""
    from diffdock_pocket_amd.batch import collate, set_time
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    g = make_3dpf_complex(seed=0, flexible_sidechains=False, n_rec=12)
    b = collate([g, g])
    set_time(b, 0.5, 0.5, 0.25, 0.5)
    nt = b["ligand"].node_t
    assert nt["tr"] is nt["rot"] is nt["sc_tor"] and nt["tor"] is not nt["tr"]
    assert nt["tr"].shape == (b["ligand"].num_nodes,) and float(nt["tr"][0]) == 0.5 and float(nt["tor"][-1]) == 0.25
    assert b.complex_t["tr"].shape == (2,) and torch.equal(b.complex_t["tor"], torch.full((2,), 0.25))
    set_time(b, torch.tensor(0.75), 0.5, 0.5, 0.5)          # a tensor-valued time is multiplied through as before
    assert torch.equal(b["atom"].node_t["tr"], torch.full((b["atom"].num_nodes,), 0.75))


def test_h2_operand_planes_layout_and_precision():
    """packing._pack_tiles_h2: fc weights as the fp16 hi/lo operand planes of v_mfma_f32_32x32x16_f16 (include/ddp_hip.h,
    ddp_conv_task_t::w2h): [tile][ks][plane][hh][column j][8 halves k = 16 ks + 8 hh + i], K zero-padded; hi + lo / 2048 carries
    22 significant bits of every weight (tiny and large ones alike)."""
    from diffdock_pocket_amd import packing as P
    g = torch.Generator().manual_seed(0)
    W = torch.randn(64, 180, generator=g) * 0.1
    W[0, :8] = torch.tensor([1e-6, -3e-5, 6e-5, 1.0, -7.3, 100.0, 0.0, 2.0 ** -14])
    pl = P._pack_tiles_h2(W, 12)
    assert pl.dtype == torch.float16 and pl.numel() == 2 * 64 * 192
    pl = pl.reshape(2, 12, 2, 2, 32, 8).float()                       # [tile, ks, plane, hh, j, i]
    rec = (pl[:, :, 0] + pl[:, :, 1] / 2048.0).permute(0, 3, 1, 2, 4).reshape(64, 192)     # [tile, j, ks, hh, i] -> [col, k]
    assert float(rec[:, 180:].abs().max()) == 0.0
    err = (rec[:, :180].double() - W.double()).abs()
    assert float((err / W.double().abs().clamp(min=2.0 ** -14)).max()) < 2.0 ** -21
    spec = P.faster_tp_spec(P.irreps_muls(60, 10, 3), P.irreps_muls(60, 10, 4), 180)
    assert P.h2_steps(spec) == 12 and P.h2_steps(P.faster_tp_spec(P.irreps_muls(60, 10, 6), (0, 2, 2, 0), 120)) == 0
    w2 = torch.randn(spec.weight_numel, 180, generator=g) * 0.05
    assert P.pack_fc2_h2(spec, w2).numel() == spec.ntiles * 2 * 12 * 64 * 8


def test_rows_kernel_unified_planes_of_the_weight_stream_and_g():
    """ddp_conv_rows' operands (include/ddp_hip.h DDP_ROWS_S*): UNIFIED planes V = S v = hi + lo, lo = fp16(V - hi) at hi's scale.
    packing.rows_stream: the fc.0 tiles of the stream carry 256 w1 in natural k order, the fc.3 tiles 256 w2 (block scale folded in) with
    k permuted by rows_kperm, the bias words sit at the scale of their tile's accumulator (fc.0: 256 x 16, fc.3: 16 x 256);
    packing.factor_weights_gh: the G columns of stage A's right-hand sides carry 32, the Gb columns 16 x 32.  A weight beyond the planes'
    range (|w| > 255) is refused."""
    from diffdock_pocket_amd import packing as P
    g = torch.Generator().manual_seed(1)
    ns, nv, layer = 60, 10, 3
    spec_g = P.faster_tp_spec(P.irreps_muls(ns, nv, layer), P.irreps_muls(ns, nv, layer + 1), 3 * ns, factorized=True)
    assert P.rows_supported(spec_g) and P.h2_steps(spec_g) == 12
    hid = 3 * ns
    w1, b1 = torch.randn(hid, hid, generator=g) * 0.07, torch.randn(hid, generator=g) * 0.1
    w2, b2 = torch.randn(spec_g.weight_numel, hid, generator=g) * 0.07, torch.randn(spec_g.weight_numel, generator=g) * 0.1
    w1[3, 5], w2[7, 11] = 1e-5, 100.0                                       # a tiny and a large weight
    wsh, bsp = P.rows_stream(spec_g, w1, b1, w2, b2)
    nts = spec_g.nct1 + sum(len(t) for _, _, t in P.rows_segments(spec_g))
    assert wsh.dtype == torch.float16 and wsh.numel() == nts * 2 * 12 * 64 * 8 and tuple(bsp.shape) == (nts, 32)
    tiles = wsh.reshape(nts, 12, 2, 2, 32, 8).float()                       # [tile, ks, plane, hh, j, i]
    rec = (tiles[:, :, 0] + tiles[:, :, 1]).permute(0, 3, 1, 2, 4).reshape(nts, 32, 192) / P.ROWS_SW     # [tile, column j, k]
    # fc.0: tile ct, column j = h column 32 ct + j, natural k
    W1 = torch.zeros(spec_g.nct1 * 32, 192)
    W1[:hid, :hid] = w1
    err1 = (rec[:spec_g.nct1].reshape(-1, 192).double() - W1.double()).abs()
    assert float((err1 / W1.double().abs().clamp(min=0.125 / P.ROWS_SW)).max()) < 2.0 ** -20      # 22 bits above |V| = 0.125, an absolute 2^-25 / S below
    assert torch.allclose(bsp[:spec_g.nct1].reshape(-1)[:hid], b1 * (P.ROWS_SW * P.ROWS_SX))
    # fc.3: the first stream tile of the first segment = packed tile `order[0]`, k permuted
    segs = P.rows_segments(spec_g)
    first = next(t for _, _, t in segs if t)[0]
    blk = next(b for b in spec_g.blocks if b.tile0 <= first < b.tile0 + max(b.ntiles, 1))
    rows = blk.column_rows().reshape(-1, 32)[first - blk.tile0]
    kp = P.rows_kperm(12)
    for j in (0, 5, 31):
        if int(rows[j]) < 0:
            continue
        want = torch.zeros(192)
        want[:hid] = w2[int(rows[j])] * blk.scale
        got = rec[spec_g.nct1, j]
        assert float((got.double() - want[kp].double()).abs().max()) < 2.0 ** -20 * max(1.0, float(want.abs().max()))
        assert abs(float(bsp[spec_g.nct1, j]) - float(b2[int(rows[j])]) * blk.scale * P.ROWS_SH * P.ROWS_SW) < 1e-3
    # stage A's right-hand sides for the plane form
    wgh, offs, widths = P.factor_weights_gh(spec_g, w2, b2)
    wg, bg, _ = P.factor_weights(spec_g, w2, b2)
    for slot in (0, 1):
        if wgh[slot] is None:
            continue
        gcp, n8 = sum(widths[slot]), (hid + 7) // 8
        assert wgh[slot].shape[1] == P.gh_ld(hid, gcp)
        # Gb columns: the fp32 form's bias part x 16 x 32
        gb = wgh[slot][:, 8 * n8 * gcp:8 * n8 * gcp + gcp]
        cum = 0
        for (_, _, c0, w, wp, _cum) in P.gh_parts(spec_g, slot):
            assert torch.allclose(gb[:, cum:cum + w], bg[slot][:, c0:c0 + w] * (P.ROWS_SH * P.ROWS_SG))
            cum += wp
        # G columns: element (k8 = 0, column 0 of the slot's first part, i) of the tile = 32 x weight[(u, column)][h column kperm[i]] x scale
        b0 = next(b for b in spec_g.blocks if b.g_slot == slot)
        for u in (0, 7):
            for i in (0, 3):
                want = float(w2[b0.w_off + (b0.g_u0 + u) * b0.n + 0, int(kp[i])]) * b0.scale * P.ROWS_SG
                assert abs(float(wgh[slot][u, i]) - want) <= 1e-6 * max(1.0, abs(want))
    w2_big = w2.clone()
    w2_big[int(rows[0]), 0] = 300.0 / blk.scale          # (a weight of the stream's first tile)
    with pytest.raises(NotImplementedError):
        P.rows_stream(spec_g, w1, b1, w2_big, b2)


def test_lazy_stats_fresh_view_rereads_the_count_block():
    """LazyStats memoises its first read, but the count block of a CAPTURED forward is rewritten by every replay -
    Sampler installs `stats.fresh()` after each replay; a fresh view reads the block again."""
    from diffdock_pocket_amd.engine import LazyStats

    class Block:
        def __init__(self):
            self.v = {"ll": 5}

        def values(self):
            return dict(self.v)

    blk = Block()
    st = LazyStats({"B": 2}, blk, {"E_ll": "ll"})
    assert dict(st) == {"B": 2, "E_ll": 5}
    blk.v["ll"] = 9                                  # (a replay rewrote the block)
    assert st["E_ll"] == 5                           # memoised
    assert st.fresh()["E_ll"] == 9 and st.fresh()["B"] == 2


def test_smooth_edge_weight_matches_the_oracle_and_tolerates_unfilled_list_entries():
    """engine.ForwardEngine._smooth_weight = get_edge_weight of the reference (all_atom_score_model.py:438-442, smooth_edges) as the
    oracle restates it (pinned by the reference-generated goldens smooth_dyn / smooth_fixed), for a fixed and a per-edge max_norm;
    entries behind a list's device-side count hold arbitrary indices and must neither fault nor change the valid rows."""
    from diffdock_pocket_amd.engine import ForwardEngine
    from oracle.ref_model import OracleScoreModel
    g = torch.Generator().manual_seed(3)
    pa, pb = torch.randn(40, 3, generator=g) * 4, torch.randn(25, 3, generator=g) * 4
    ia = torch.randint(0, 40, (200,), generator=g).to(torch.int32)
    ib = torch.randint(0, 25, (200,), generator=g).to(torch.int32)
    orc = OracleScoreModel.__new__(OracleScoreModel)
    orc.cfg = OracleConfig(smooth_edges=True)
    vec = pb[ib.long()] - pa[ia.long()]
    for mx in (5.0, torch.rand(200, generator=g) * 8 + 2):
        want = orc._edge_weight(vec, mx).squeeze(-1)
        got = ForwardEngine._smooth_weight(pa, ia, pb, ib, mx)
        assert torch.equal(got, want)
        assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    ia2, ib2 = ia.clone(), ib.clone()
    ia2[150:], ib2[150:] = 10 ** 6, -7          # "unfilled" capacity entries
    got = ForwardEngine._smooth_weight(pa, ia2, pb, ib2, 5.0)
    assert torch.equal(got[:150], ForwardEngine._smooth_weight(pa, ia, pb, ib, 5.0)[:150]) and torch.isfinite(got).all()
    orc.cfg = OracleConfig(smooth_edges=False)
    assert orc._edge_weight(vec, 5.0) == 1.0


@pytest.mark.parametrize("fmt", [0, 1])
def test_g_plane_forms_host_side_against_the_decoder(fmt):
    """The two plane forms of a factorised conv's G (ddp_conv_task_t::gh_fmt; include/ddp_hip.h): packing.factor_weights_gh orders the
    product's columns, packing.gh_dest_table says where the two pieces of every 8-column group go, the kernel's drain is one conversion
    per group.  Emulated here on the CPU exactly as csrc/ddp_gemm.hip drains (form 0: a plane group = 8 fp16 hi words + 8 fp16 lo words at the
    table's two places, any other group its 8 fp32 values; form 1: group g at byte 24 g - a plane group = the fp32 pattern + 0x10 truncated to
    fp16 and its mantissa bits 12 .. 5, any other group the six fp32 values of product columns 0, 1, 4, 5, 2, 6), and read back with
    tests/helpers.decode_gh_rows, which is written from the header's description of the BYTES alone: the decoded planes must be the fp64
    product (to the forms' precision), the Gb columns exact, and the row lengths DDP_GH_LD / DDP_GH3_LD."""
    import numpy as np
    from diffdock_pocket_amd import packing as P
    from helpers import decode_gh_rows
    g = torch.Generator().manual_seed(5 + fmt)
    ns, nv, layer = 60, 10, 3
    spec_g = P.faster_tp_spec(P.irreps_muls(ns, nv, layer), P.irreps_muls(ns, nv, layer + 1), 3 * ns, factorized=True)
    hid = 3 * ns
    w2, b2 = torch.randn(spec_g.weight_numel, hid, generator=g) * 0.07, torch.randn(spec_g.weight_numel, generator=g) * 0.1
    wgh, offs, widths = P.factor_weights_gh(spec_g, w2, b2, fmt=fmt)
    wg0, _, _ = P.factor_weights_gh(spec_g, w2, b2, fmt=0)
    n8 = (hid + 7) // 8
    for slot in (0, 1):
        if wgh[slot] is None:
            continue
        W = wgh[slot]
        gcp = sum(widths[slot])
        ncols = W.shape[1]
        ld = P.gh3_ld(hid, gcp) if fmt == 1 else P.gh_ld(hid, gcp)
        assert ncols % 32 == 0 and (ncols == ld if fmt != 1 else ncols % 128 == 0 and ld == 6 * ncols // 8 and (ld * 4) % 384 == 0)
        tab = P.gh_dest_table(widths[slot], n8, ncols, fmt=fmt)
        x = torch.randn(7, W.shape[0], generator=g)
        prod = (x.double() @ W.double()).float()                              # what the block's accumulator holds (fp32)
        raw = np.zeros((7, ld * 4), dtype=np.uint8)
        for gi in range(ncols // 8):
            v = prod[:, 8 * gi:8 * gi + 8]
            o0, o1, plane = int(tab[gi, 0]) & ~3, int(tab[gi, 1]), int(tab[gi, 0]) & 1
            if fmt == 1:
                assert int(tab[gi, 0]) & ~1 == 6 * gi
                if plane:
                    b = v.contiguous().numpy().view(np.uint32) + np.uint32(0x10)
                    sgn, e, m = (b >> 31).astype(np.int64), ((b >> 23) & 0xff).astype(np.int64), (b & 0x7fffff).astype(np.int64)
                    e16 = e - 112
                    hw = np.where(e16 >= 1, (e16 << 10) | (m >> 13), (m | 0x800000) >> np.minimum(14 - e16, 40))     # (truncation, subnormal results too)
                    hw = np.where(e == 0, 0, hw) | (sgn << 15)
                    raw[:, 24 * gi:24 * gi + 16] = hw.astype(np.uint16).view(np.uint8).reshape(7, 16)
                    raw[:, 24 * gi + 16:24 * gi + 24] = ((b >> 5) & 0xff).astype(np.uint8)
                else:
                    six = v[:, list(P.GH3_FP32_COLS)].contiguous().numpy()
                    raw[:, 24 * gi:24 * gi + 24] = six.view(np.uint8).reshape(7, 24)
            elif plane:
                hi = v.to(torch.float16)
                rest = v - hi.float()
                raw[:, 4 * o0:4 * o0 + 16] = hi.numpy().view(np.uint8).reshape(7, 16)
                raw[:, 4 * o1:4 * o1 + 16] = rest.to(torch.float16).numpy().view(np.uint8).reshape(7, 16)
            else:
                raw[:, 4 * o0:4 * o0 + 16] = v[:, :4].contiguous().numpy().view(np.uint8).reshape(7, 16)
                raw[:, 4 * o1:4 * o1 + 16] = v[:, 4:8].contiguous().numpy().view(np.uint8).reshape(7, 16)
        rows = torch.from_numpy(raw.view(np.float32).copy())
        V, Gb = decode_gh_rows(rows, widths[slot], hid, fmt)
        # the same planes and bias columns as form 0's right-hand side gives (its column order is the documented one)
        W0 = wg0[slot]
        want = (x.double() @ W0.double())
        wantV = torch.zeros(7, n8, gcp, 8, dtype=torch.float64)
        cum = 0
        for w in widths[slot]:
            wantV[:, :, cum:cum + w] = want[:, 8 * n8 * cum:8 * n8 * (cum + w)].reshape(7, n8, w, 8)
            cum += w
        wantB = want[:, 8 * n8 * gcp:8 * n8 * gcp + gcp]
        tolV = 2.0 ** -21 * wantV.abs() + 2.0 ** -24 if fmt != 1 else 2.0 ** -19 * wantV.abs() + 2.0 ** -24
        assert bool(((V - wantV).abs() <= tolV + 1e-6 * wantV.abs()).all()), (fmt, slot, float((V - wantV).abs().max()))
        # (the Gb columns of the padded parts only: the others are zero on both sides)
        assert bool(((Gb - wantB).abs() <= 1e-6 * wantB.abs() + 1e-9).all()), (fmt, slot)


def test_rows_kernel_operand_images_of_the_16x16x32_form():
    """ddp_conv_task_t::rows_form = 1 (csrc/ddp_conv_rows16.hip, include/ddp_hip.h): the weight stream in the operand images of
    v_mfma_f32_16x16x32_f16 - per 32-column tile 2 NS fragments [k32 step s][column tile ct][plane] of [k group g][column n][8 halves] =
    plane(256 w)[column 16 ct + n][k = 32 s + 8 g + i] - with the k of the fc.3 tiles in NATURAL order, and fc.0's output columns placed
    inside every 32-column tile at DDP_ROWS16_POS (h column 32 t + 8 g + i at position 16 (i / 4) + 4 g + i % 4, bias words with them): the
    transposed fc1 product then leaves a lane's accumulator registers as its A fragment of the later products in natural k order.  The
    stage-A right-hand sides keep form 0's columns with the k's of every 8-group un-permuted."""
    from diffdock_pocket_amd import packing as P
    g = torch.Generator().manual_seed(2)
    ns, nv, layer = 60, 10, 3
    spec_g = P.faster_tp_spec(P.irreps_muls(ns, nv, layer), P.irreps_muls(ns, nv, layer + 1), 3 * ns, factorized=True)
    hid = 3 * ns
    w1, b1 = torch.randn(hid, hid, generator=g) * 0.07, torch.randn(hid, generator=g) * 0.1
    w2, b2 = torch.randn(spec_g.weight_numel, hid, generator=g) * 0.07, torch.randn(spec_g.weight_numel, generator=g) * 0.1
    wsh, bsp = P.rows_stream(spec_g, w1, b1, w2, b2, form=1)
    wsh0, bsp0 = P.rows_stream(spec_g, w1, b1, w2, b2, form=0)
    assert wsh.shape == wsh0.shape and bsp.shape == bsp0.shape
    nts = bsp.shape[0]
    tiles = wsh.reshape(nts, 6, 2, 2, 4, 16, 8).float()                     # [tile, s, ct, plane, g, n, i]
    rec = (tiles[:, :, :, 0] + tiles[:, :, :, 1]) / P.ROWS_SW               # [tile, s, ct, g, n, i]
    rec = rec.permute(0, 2, 4, 1, 3, 5).reshape(nts, 32, 192)               # [tile, column 16 ct + n, k = 32 s + 8 g + i]
    pos = P.rows16_pos(192)
    assert sorted(pos.tolist()) == list(range(192)) and int(pos[8 * 1 + 5]) == 16 * 1 + 4 * 1 + 1      # (g = 1, i = 5 -> position 21)
    # fc.0: position pos[c] of the stream holds h column c (natural k = edge_attr_ column)
    W1 = torch.zeros(192, 192)
    W1[:hid, :hid] = w1
    got1 = rec[:spec_g.nct1].reshape(192, 192)
    assert float((got1[pos] - W1).abs().max()) < 2.0 ** -20
    bias1 = torch.zeros(192)
    bias1[:hid] = b1 * (P.ROWS_SW * P.ROWS_SX)
    assert torch.allclose(bsp[:spec_g.nct1].reshape(-1)[pos], bias1)
    # fc.3: the same tiles as form 0 in the same stream order, natural k: form 0's tile is this one with its k permuted by rows_kperm
    t0 = wsh0.reshape(nts, 12, 2, 2, 32, 8).float()                          # [tile, ks, plane, hh, j, i]
    rec0 = ((t0[:, :, 0] + t0[:, :, 1]).permute(0, 3, 1, 2, 4).reshape(nts, 32, 192)) / P.ROWS_SW
    kp = P.rows_kperm(12)
    nat0 = torch.zeros_like(rec0)
    nat0[:, :, kp] = rec0                                                    # slot -> natural k
    assert float((rec[spec_g.nct1:] - nat0[spec_g.nct1:]).abs().max()) < 2.0 ** -20
    assert torch.equal(bsp[spec_g.nct1:], bsp0[spec_g.nct1:])
    # stage A's right-hand sides: identity k order inside the groups
    wg1, _, widths1 = P.factor_weights_gh(spec_g, w2, b2, form=1)
    wg0, _, widths0 = P.factor_weights_gh(spec_g, w2, b2, form=0)
    assert widths1 == widths0
    n8 = (hid + 7) // 8
    for slot in (0, 1):
        if wg1[slot] is None:
            continue
        gcp = sum(widths1[slot])
        a = wg1[slot][:, :8 * n8 * gcp]
        b = wg0[slot][:, :8 * n8 * gcp]
        assert torch.equal(wg1[slot][:, 8 * n8 * gcp:], wg0[slot][:, 8 * n8 * gcp:])     # Gb columns: the same
        cum = 0
        for w in widths1[slot]:
            A = a[:, 8 * n8 * cum:8 * n8 * (cum + w)].reshape(-1, n8, w, 8).permute(0, 2, 1, 3).reshape(-1, w, 8 * n8)      # [u, column, k slot]
            B = b[:, 8 * n8 * cum:8 * n8 * (cum + w)].reshape(-1, n8, w, 8).permute(0, 2, 1, 3).reshape(-1, w, 8 * n8)
            nat = torch.zeros(A.shape[0], w, 192)
            nat[:, :, kp[:8 * n8]] = B                                        # form 0: slot -> natural k
            assert torch.equal(A, nat[:, :, :8 * n8])
            cum += w


def test_rows_kernel_options_are_consistent(monkeypatch):
    """model.rows_mfma16 (the row-stationary conv kernel on v_mfma_f32_16x16x32_f16) and model.g_planes3 (G in three bytes per value, 19
    significant bits): independent of each other since ABI 17 (both kernels read both plane forms), both setters drop the packed weights
    (epoch) and tell the conv layers; the environment sets the defaults."""
    from diffdock_pocket_amd.score_model import TensorProductConvLayer, TensorProductScoreModel
    from oracle.cases import CASES
    case = CASES["cfg2_small"]
    kw = dict(case.model_kwargs())
    kw.update(case.ctor_extras())
    kw["device"] = torch.device("cpu")
    monkeypatch.delenv("DDP_ROWS_MFMA16", raising=False)
    monkeypatch.delenv("DDP_G_PLANES3", raising=False)
    m = TensorProductScoreModel(**kw)
    convs = [c for c in m.modules() if isinstance(c, TensorProductConvLayer)]
    from diffdock_pocket_amd.score_model import G_PLANES3_DEFAULT, ROWS_MFMA16_DEFAULT
    d3, d16 = G_PLANES3_DEFAULT == "1", ROWS_MFMA16_DEFAULT == "1"
    assert m.rows_mfma16 == d16 and m.g_planes3 == d3
    assert all(getattr(c, "rows_form", 0) == int(d16) and getattr(c, "gh_fmt", 0) == int(d3) for c in convs)
    e0 = m.__dict__.get("_packed_epoch", 0)
    m.g_planes3 = not d3
    assert m.g_planes3 != d3 and m.rows_mfma16 == d16 and all(c.rows_form == int(d16) and c.gh_fmt == int(not d3) for c in convs)
    e1 = m.__dict__["_packed_epoch"]
    assert e1 > e0
    m.rows_mfma16 = not d16
    assert m.__dict__["_packed_epoch"] > e1 and all(c.rows_form == int(not d16) and c.gh_fmt == int(not d3) for c in convs)
    m.g_planes3 = d3
    m.rows_mfma16 = d16
    assert all(c.rows_form == int(d16) and c.gh_fmt == int(d3) for c in convs)
    for v3 in ("0", "1"):
        for v16 in ("0", "1"):
            monkeypatch.setenv("DDP_G_PLANES3", v3)
            monkeypatch.setenv("DDP_ROWS_MFMA16", v16)
            m2 = TensorProductScoreModel(**kw)
            assert m2.g_planes3 == (v3 == "1") and m2.rows_mfma16 == (v16 == "1")


def test_direct_conv_weight_streams_of_segment_ranges():
    """ddp_conv_task_t::rows_seg0 / rows_seg1 / rows_nts (a direct conv as several tasks of output-segment ranges, each with a weight stream of
    its own): packing.rows_split_segments cuts the shape's segments into contiguous ranges that cover them once, the largest as small as
    possible; the ranges' streams (fc.0's tiles, then the range's tiles) put together are the whole stream; with bias_in_k the fc.3 bias words
    are zero, k row `hid` of every fc.3 tile holds the tile's bias and fc.0's output column `hid` is the constant 1."""
    from diffdock_pocket_amd import packing as P
    g = torch.Generator().manual_seed(3)
    ns, nv, layer = 60, 10, 3
    spec = P.faster_tp_spec(P.irreps_muls(ns, nv, layer), P.irreps_muls(ns, nv, layer + 1), 3 * ns)
    assert not spec.factorized and P.rows_bias_in_k(spec) and P.rows_supported(spec)
    hid = 3 * ns
    w1, b1 = torch.randn(hid, hid, generator=g) * 0.07, torch.randn(hid, generator=g) * 0.1
    w2, b2 = torch.randn(spec.weight_numel, hid, generator=g) * 0.07, torch.randn(spec.weight_numel, generator=g) * 0.1
    counts = [len(t) for _, _, t in P.rows_segments(spec)]
    for n in (1, 2, 3, 6, 9):
        rg = P.rows_split_segments(spec, n)
        assert len(rg) == min(n, len(counts)) and rg[0][0] == 0 and rg[-1][1] == len(counts)
        assert all(a[1] == b[0] for a, b in zip(rg[:-1], rg[1:])) and [c for _, _, c in rg] == [sum(counts[a:b]) for a, b, _ in rg]
    assert [c for _, _, c in P.rows_split_segments(spec, 2)] == [167, 167]
    whole, bs = P.rows_stream(spec, w1, b1, w2, b2, form=1, bias_in_k=True)
    tile = 2 * 12 * 1024 // 2                      # halves per tile (NS = 12)
    whole = whole.reshape(-1, tile)
    n1 = spec.nct1
    assert whole.shape[0] == n1 + sum(counts) and bs.shape == (n1 + sum(counts), 32)
    assert float(bs[n1:].abs().max()) == 0.0 and float(bs[:n1].abs().max()) > 0.0
    parts = []
    for a, b, c in P.rows_split_segments(spec, 3):
        w, bsp = P.rows_stream(spec, w1, b1, w2, b2, form=1, bias_in_k=True, seg_range=(a, b))
        w = w.reshape(-1, tile)
        assert w.shape[0] == n1 + c and torch.equal(w[:n1], whole[:n1]) and torch.equal(bsp[:n1], bs[:n1])
        parts.append(w[n1:])
    assert torch.equal(torch.cat(parts), whole[n1:])
    # the constant-1 column of fc.0: position DDP_ROWS16_POS of h column `hid` inside its tile carries the bias word 1 x ROWS_SW ROWS_SX
    t_, j_ = hid // 32, hid % 32
    pos = 16 * ((j_ & 7) >> 2) + 4 * (j_ >> 3) + (j_ & 3)
    assert float(bs[t_, pos]) == P.ROWS_SW * P.ROWS_SX
