"""GPU parity: the HIP path (through the C ABI of libddp_hip.so) against the CPU oracle on identical seeded inputs
and against the golden vectors captured from the reference's own model files.

Tolerance: BASELINE.json north_star asks for 1e-4 relative in fp32; we assert 1e-4 of the largest component of each
output vector (scores are 3-vectors / per-bond scalars whose small components carry no more absolute precision
than the large ones)."""
import os

import pytest
import torch

from oracle import thirdparty as tp
from oracle.cases import CASES
from oracle.ref_model import OracleConfig, OracleScoreModel, gaussian_smearing

from helpers import case_inputs, conv_stats_excess, elementwise_excess, rel_err, rowwise_excess

pytestmark = pytest.mark.gpu
TOL = 1e-4
ATOL_FRAC = 1e-6   # absolute floor of the element-wise check, as a fraction of the array's largest |score|


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _model_for(case, sd):
    from diffdock_pocket_amd.score_model import TensorProductScoreModel
    kw = dict(case.model_kwargs())
    kw.update(case.ctor_extras())
    kw["device"] = _dev()
    model = TensorProductScoreModel(**kw)
    model.load_state_dict(sd, strict=True)
    return model.to(_dev()).eval()


@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_oracle_and_golden(name):
    case, gold, batch, sd = case_inputs(name)
    oracle = OracleScoreModel(case.oracle_config(), sd)
    want = oracle(case.make_batch())
    model = _model_for(case, sd)
    got = model(batch.to(_dev()))
    torch.cuda.synchronize()
    keys = ("tr", "rot", "tor", "sc_tor")
    if case.confidence_mode:   # the confidence model reuses the convs, scalar-mean + MLP head
        got, want, keys = (got,), (want,), ("confidence",)
    for g, w, k in zip(got, want, keys):
        g = g.float().cpu()
        assert g.shape == w.shape, (k, g.shape, w.shape)
        assert rel_err(g, w) < TOL, (name, k, "vs oracle", rel_err(g, w))
        assert rel_err(g, gold["outputs"][k]) < TOL, (name, k, "vs reference golden", rel_err(g, gold["outputs"][k]))
        assert torch.isfinite(g).all()
        if k in ("tor", "sc_tor"):   # per-bond arrays: every element on its own, |d| <= 1e-4 |ref| + ATOL_FRAC max|ref|
            assert elementwise_excess(g, w, TOL, ATOL_FRAC) <= 1.0, (name, k, "element-wise vs oracle", elementwise_excess(g, w, TOL, ATOL_FRAC))
            assert elementwise_excess(g, gold["outputs"][k], TOL, ATOL_FRAC) <= 1.0, (name, k, "element-wise vs golden")
        if k in ("tr", "rot"):       # per graph: |d_row|_inf <= 1e-4 |ref_row|_inf + ATOL_FRAC max|ref|
            assert rowwise_excess(g, w, TOL, ATOL_FRAC) <= 1.0, (name, k, "row-wise vs oracle", rowwise_excess(g, w, TOL, ATOL_FRAC))
            assert rowwise_excess(g, gold["outputs"][k], TOL, ATOL_FRAC) <= 1.0, (name, k, "row-wise vs golden")
    st = model.last_stats
    assert st["E_aa"] == gold["edge_counts"]["aa"]
    assert st["E_lr"] == int(oracle.record["lr"].shape[1]) and st["E_la"] == int(oracle.record["la"].shape[1])
    assert st["E_ll"] == int(oracle.record["ll"].shape[1])


@pytest.mark.parametrize("name", ["cfg1_full", "cfg2_noflex"])
def test_forward_accepts_a_pyg_shaped_batch(name):
    """The drop-in is handed a PyG HeteroDataBatch by utils/sampling.py:112-120.  torch_geometric is not installed, so the
    forward runs on helpers.PyGLikeBatch - canonical 3-tuple edge keys, stores created on access, `in` by attribute name,
    len(store) - and must give bit-for-bit the result of the HeteroBatch path, and leave the documented side effects
    (node_sigma_emb, data['atom','atom'].edge_index, graph_sigma_emb; all_atom_score_model.py:369-373,447-454,530)."""
    from helpers import PyGLikeBatch
    case, gold, batch, sd = case_inputs(name)
    dev = _dev()
    model = _model_for(case, sd)
    want = [t.clone() for t in model(case.make_batch().to(dev))]
    duck = PyGLikeBatch.from_hetero_batch(case.make_batch(), dev)
    if not case.flexible_sidechains:     # utils/sampling.py:84 deletes the store when no side chain is flexible
        if "flexResidues" in duck._node:
            del duck["flexResidues"]
    got = model(duck)
    for g, w, k in zip(got, want, ("tr", "rot", "tor", "sc_tor")):
        assert torch.equal(g, w), k
        assert rel_err(g.float().cpu(), gold["outputs"][k]) < TOL, (name, k)
    assert duck["ligand"].node_sigma_emb.shape[0] == duck["ligand"].pos.shape[0]
    assert duck["atom", "atom"].edge_index.shape[1] == gold["edge_counts"]["aa"]
    assert duck.graph_sigma_emb.shape[0] == duck.num_graphs


@pytest.mark.parametrize("flex", [False, True])
def test_bench_batch_samples_match_oracle(flex):
    """BASELINE configs[1] (rigid receptor) / configs[2] (flexible side chains) exactly as bench.py runs them: the 40-sample
    batch of the full 3dpf complex through the cfg2 model (ns=60 nv=10 L=6), built by bench.build_model / the sampler with
    bench.py's seeds.  One HIP forward of the whole batch at the first schedule position and one after ten denoising
    steps (mid schedule: other cutoffs, other edge sets, moved side chains).  Rigid receptor, first position: ALL 40 samples
    are compared with the CPU oracle (8-graph oracle batches); the other three forwards: eight samples each (reference
    semantics: a graph's scores do not depend on its batch mates)."""
    import bench
    from diffdock_pocket_amd.batch import collate, set_time
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    model, kw = bench.build_model("cfg2", flex, dev)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    oracle = OracleScoreModel(OracleConfig(ns=kw["ns"], nv=kw["nv"], num_conv_layers=kw["num_conv_layers"],
                                           sigma_embed_dim=kw["sigma_embed_dim"], distance_embed_dim=kw["distance_embed_dim"],
                                           cross_distance_embed_dim=kw["cross_distance_embed_dim"],
                                           flexible_sidechains=flex, embedding_scale=1000.0), sd)
    g = make_3dpf_complex(seed=0, flexible_sidechains=flex)
    smp = Sampler(model, g, 40, dev, SamplerConfig(inference_steps=20, flexible_sidechains=flex), seed=0)
    smp.randomize()
    sched = get_t_schedule(20)
    for t_idx in (0, 10):
        # rigid receptor, first schedule position: ALL 40 samples (five 8-graph oracle forwards, ~90 s of CPU); otherwise 8 picks
        picks = list(range(40)) if (not flex and t_idx == 0) else [0, 5, 11, 17, 23, 29, 34, 39]
        while getattr(smp, "_steps_done", 0) < t_idx:
            smp.step(getattr(smp, "_steps_done", 0), sched)
            smp._steps_done = getattr(smp, "_steps_done", 0) + 1
        t = float(sched[t_idx])
        got = [o.float().cpu() for o in smp.scores(t)]
        assert got[0].shape == (40, 3) and model.last_stats["B"] == 40
        T, S_ = got[2].numel() // 40, got[3].numel() // 40
        for c0 in range(0, len(picks), 8):
            part = picks[c0:c0 + 8]
            graphs = []
            for i in part:
                c = g.clone()
                c["ligand"].pos, c["atom"].pos = smp.lig_pos[i].cpu().clone(), smp.atom_pos[i].cpu().clone()
                graphs.append(c)
            cb = collate(graphs)
            set_time(cb, t, t, t, t)
            want = oracle(cb)
            sel = [got[0][part], got[1][part], got[2].reshape(40, T)[part].reshape(-1), got[3].reshape(40, S_)[part].reshape(-1)]
            for a, w, k in zip(sel, want, ("tr", "rot", "tor", "sc_tor")):
                assert a.shape == w.shape, (k, a.shape, w.shape)
                assert (k == "sc_tor" and not flex) or w.numel() > 0
                assert rel_err(a, w) < TOL, (flex, t_idx, part, k, rel_err(a, w))
                if k in ("tor", "sc_tor"):
                    assert elementwise_excess(a, w, TOL, ATOL_FRAC) <= 1.0, (flex, t_idx, part, k, elementwise_excess(a, w, TOL, ATOL_FRAC))
                if k in ("tr", "rot"):
                    assert rowwise_excess(a, w, TOL, ATOL_FRAC) <= 1.0, (flex, t_idx, part, k, rowwise_excess(a, w, TOL, ATOL_FRAC))
    smp.close()     # (the captured 40-sample step holds ~40 GB of device memory)


@pytest.mark.parametrize("name", ["cfg2_full_noflex", "cfg2_full_flex", "cfg1_full", "ns24_l3", "hetero_cfg1", "hetero_cfg2", "small32_readme"])
def test_every_conv_output_matches_the_reference_hooks(name):
    """The goldens hold, for every conv call of the reference's forward (forward hooks on conv_layers[0..9L), final_conv,
    tor_bond_conv, sc_tor_bond_conv: oracle/make_golden.py), the output shape, mean|.| and a 64-point strided sample.  The HIP
    model's debug hook returns the same tensors (segmented mean + BatchNorm of each conv's messages alone); parity at the
    four outputs alone could hide a wrong block behind the BatchNorm-ed residual sums."""
    case, gold, batch, sd = case_inputs(name)
    model = _model_for(case, sd)
    model.debug_conv_outputs = {}
    model(batch.to(_dev()))
    torch.cuda.synchronize()
    stats = gold["conv_stats"]
    assert set(model.debug_conv_outputs) == set(stats), (sorted(set(stats) ^ set(model.debug_conv_outputs)))
    worst = {}
    for key, st in stats.items():
        got = model.debug_conv_outputs[key]
        if list(st["shape"]) == []:      # an empty edge set: the reference returns the scalar 0 (models/score_model.py:109-111)
            assert float(got.abs().max()) == 0.0, key
            continue
        w, m = conv_stats_excess(got, st, rtol=TOL, atol_frac=TOL)
        worst[key] = (w, m)
    bad = {k: v for k, v in worst.items() if v[0] > 1.0 or v[1] > TOL}
    assert not bad, (name, bad)


@pytest.mark.parametrize("name", ["cfg2_small", "cfg1_edge", "ns24_l3", "cfg2_full_noflex", "hetero_cfg1"])
def test_node_encoders_and_sigma_tables_match_oracle(name):
    """AtomEncoder / OldAtomEncoder (models/score_model.py:54-82, :17-52), the sinusoidal sigma
    embedding (utils/diffusion_utils.py:73-84) and the node-dependent part of the edge-embedding MLPs' first Linear, all from
    ONE ddp_node_linear launch (csrc/ddp_node.hip, fp32 MFMA), against the oracle's restatement on the same batch.
    cfg1_edge is the legacy encoder with an ESM block (two stages, literal legacy column slicing)."""
    from diffdock_pocket_amd.synthetic import LIG_FEATURE_DIMS, REC_ATOM_FEATURE_DIMS, REC_RESIDUE_FEATURE_DIMS
    case, gold, batch, sd = case_inputs(name)
    dev = _dev()
    oracle = OracleScoreModel(case.oracle_config(), sd)
    model = _model_for(case, sd)
    b = batch.to(dev)
    xl, xr, xa, pre = model._node_tables(b["ligand"], b["receptor"], b["atom"], dev)
    torch.cuda.synchronize()
    cb = case.make_batch()
    ns, sdim = model.ns, model.sigma_embed_dim
    embs = {}
    for nt, prefix, ncat, x in (("ligand", "lig_node_embedding", len(LIG_FEATURE_DIMS), xl),
                                ("receptor", "rec_node_embedding", len(REC_RESIDUE_FEATURE_DIMS), xr),
                                ("atom", "atom_node_embedding", len(REC_ATOM_FEATURE_DIMS), xa)):
        st = cb[nt]
        emb = embs[nt] = oracle._emb(st.node_t["tr"])
        want = oracle._atom_encoder(prefix, torch.cat([st.x.float(), emb], 1), ncat)
        got = x.cpu()
        assert got.shape == (st.x.shape[0], model._ldx)
        assert rel_err(got[:, :ns], want) < 2e-5, (nt, rel_err(got[:, :ns], want))
        assert float(got[:, ns:].abs().max()) == 0.0                       # the irreps of later layers start at zero
        assert float((b[nt].node_sigma_emb.cpu() - emb).abs().max()) < 2e-6, nt   # the side effect the reference leaves
    nf, dd = model.in_lig_edge_features, model.distance_embed_dim
    for key, mlp, nt, s0 in (("ll", "lig_edge_embedding", "ligand", nf), ("lr", "lr_edge_embedding", "ligand", 0),
                             ("la", "la_edge_embedding", "ligand", 0), ("rr", "rec_edge_embedding", "receptor", 0),
                             ("aa", "atom_edge_embedding", "atom", 0), ("ar", "ar_edge_embedding", "atom", 0),
                             ("center", "center_edge_embedding", "ligand", dd)):
        W, bias = sd[mlp + ".0.weight"], sd[mlp + ".0.bias"]
        want = embs[nt].double() @ W[:, s0:s0 + sdim].t().double() + bias.double()
        assert rel_err(pre[key].cpu(), want.float()) < 2e-5, (key, rel_err(pre[key].cpu(), want.float()))


@pytest.mark.parametrize("name", ["cfg2_small", "cfg1_full", "cfg1_edge", "cfg2_noflex", "conf_ns24_l5", "hetero_cfg1", "hetero_cfg2", "hetero_conf"])
def test_results_do_not_depend_on_list_capacities(name):
    """The pose-dependent lists are sized for the worst case and their counts live on the device (engine.py).  With
    `exact_sizes` every count is read back and every list cut to its length, i.e. all kernels run with host-known sizes and
    exact grids - the host-driven form of round 2.  Both must give the same bits."""
    case, gold, batch, sd = case_inputs(name)
    model = _model_for(case, sd)
    b = case.make_batch().to(_dev())
    a = model(b)
    a = [t.clone() for t in (a if isinstance(a, tuple) else (a,))]
    model.exact_sizes = True
    c = model(b)
    c = c if isinstance(c, tuple) else (c,)
    for x, y in zip(a, c):
        assert torch.equal(x, y)


@pytest.mark.parametrize("flex", [False, True])
def test_graph_replay_equals_launch_by_launch(flex):
    """sampler.Sampler captures its third step in a hipGraph and replays it.  Poses must be bit for bit those of the same job
    launched kernel by kernel - also when ordinary forwards on the current poses (Sampler.scores: the model's static-graph
    cache then holds entries for exactly these poses) run between the steps, before and after the capture."""
    import bench
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    sched = get_t_schedule(20)
    g = make_3dpf_complex(seed=0, flexible_sidechains=flex)
    out = {}
    for graph in (False, True):
        model, kw = bench.build_model("cfg2", flex, dev)
        smp = Sampler(model, g, 6, dev, SamplerConfig(inference_steps=20, flexible_sidechains=flex, hip_graph=graph), seed=0)
        smp.randomize()
        poses = []
        for j, i in enumerate((0, 1, 2, 10, 11, 12)):
            if j != 4:
                smp.scores(float(sched[i]))
            smp.step(i, sched)
            poses.append((smp.lig_pos.clone(), smp.atom_pos.clone()))
        assert bool(smp._graph) == graph
        out[graph] = poses
    for (l0, a0), (l1, a1) in zip(out[False], out[True]):
        assert torch.equal(l0, l1) and torch.equal(a0, a1)


def test_smooth_edges_in_the_sampler_and_under_replay():
    """smooth_edges (all_atom_score_model.py:438-442; parity: the goldens smooth_dyn / smooth_fixed in
    test_forward_matches_oracle_and_golden): the edge weights are PyTorch launches on the harmonics behind ddp_edge_featurize.  In a
    sampling batch (receptor-side sharing, forked front, captured step) the replayed steps must be bit for bit the launch-by-launch
    ones - and the option must change the scores."""
    import bench
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    sched = get_t_schedule(20)
    g = make_3dpf_complex(seed=0, flexible_sidechains=True)
    out = {}
    for smooth, graph in ((True, False), (True, True), (False, True)):
        model, kw = bench.build_model("cfg1", True, dev)
        model.smooth_edges = smooth
        smp = Sampler(model, g, 6, dev, SamplerConfig(inference_steps=20, flexible_sidechains=True, hip_graph=graph), seed=0)
        smp.randomize()
        first = [t.clone() for t in smp.scores(float(sched[0]))]
        for i in range(5):
            smp.step(i, sched)
        assert bool(smp._graph) == graph
        out[smooth, graph] = (first, smp.lig_pos.clone(), smp.atom_pos.clone(), [t.clone() for t in smp.scores(float(sched[5]))])
        smp.close()
        del smp, model
    a, b = out[True, False], out[True, True]
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for x, y in zip(a[0] + a[3], b[0] + b[3]):
        assert torch.equal(x, y)
    assert not torch.equal(out[True, True][0][0], out[False, True][0][0])


@pytest.mark.parametrize("name", ["cfg2_full_noflex", "cfg2_small", "small32_readme", "ns24_l3", "cfg1_full"])
def test_fp16_split_products_against_the_fp32_mfma_form(name):
    """Round 4: the fc products of the conv kernels run on the fp16 matrix cores with both operands split in two halves
    (v = hi + lo / 2048; three v_mfma_f32_32x32x16_f16 per 16 k, fp32 accumulation: csrc/ddp_conv.hip, "h2").  The exact fp32 MFMA
    form of rounds 1 - 3 stays in the library (model.conv_h2 = False).  Both against the fp64 oracle on the same batch: the h2
    form must be in the fp32 form's error class (the rounding noise of either form is amplified by the random-weight layers alike:
    within a factor 4 of it or 5e-6 of the largest component - the path's tolerance is 1e-4), and the two forms must agree to 1e-5 -
    every size class (ns = 60 / 32 / 24 / 16), factorised and direct kernels.  (Product by product the h2 form is the MORE accurate
    one - 0.85e-7 against 1.8e-7 of sum|a b|, profiles/r04_f16x2_mfma_micro.txt.)"""
    case, gold, batch, sd = case_inputs(name)
    want = OracleScoreModel(case.oracle_config(), sd, dtype=torch.float64)(case.make_batch())
    model = _model_for(case, sd)
    out = {}
    for h2 in (True, False):
        model.conv_h2 = h2
        got = model(case.make_batch().to(_dev()))
        out[h2] = [t.double().cpu() for t in got]
    assert model.__dict__.get("h2_recoveries", 0) == 0                            # (no forward fell back to the fp32 form)
    assert any(not torch.equal(a, b) for a, b in zip(out[True], out[False]))      # (the switch does switch)
    model.conv_h2 = True
    rep = model.split_form_error(case.make_batch().to(_dev()))                    # (the user-facing form of this comparison)
    assert rep and "recovered" not in rep and all(v < 1e-5 for v in rep.values()), rep
    for a, b, w, k in zip(out[True], out[False], want, ("tr", "rot", "tor", "sc_tor")):
        if w.numel() == 0:
            continue
        scale = float(w.abs().max())
        e_h2, e_32 = float((a - w).abs().max()) / scale, float((b - w).abs().max()) / scale
        assert e_h2 < max(4.0 * e_32, 5e-6), (name, k, e_h2, e_32)
        assert float((a - b).abs().max()) / scale < 1e-5, (name, k, float((a - b).abs().max()) / scale)


@pytest.mark.parametrize("name", ["cfg1_full", "cfg2_small"])
def test_values_outside_the_fp16_range_are_recovered_in_the_same_call(name):
    """The h2 kernels split fp32 values into two fp16 halves; a value beyond +-65504 cannot be split.  It is neither clamped nor
    silently turned into an infinity, and the caller never sees the spoiled result: the kernel raises a flag in pinned host memory
    (ddp_conv_task_t::h2_range_flag, ddp_stage_a_h2 / _gh's range_flag), `model(batch)` reads it before it returns - the reference's
    caller synchronises on the next line anyway, utils/sampling.py:122-125 - and reruns the forward in the exact fp32 MFMA form;
    `Sampler.run` (graph replay) checks once per run and reruns the job from its first step.  Default settings, no switch to know about:
    scores within 1e-4 of the oracle with conv_layers.0.fc.0.weight x 3e5 (h = relu(fc1) of one conv far beyond 65504)."""
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    case, gold, batch, sd = case_inputs(name)
    sd = dict(sd)
    sd["conv_layers.0.fc.0.weight"] = sd["conv_layers.0.fc.0.weight"] * 3e5
    if name != "cfg1_full":     # (the deeper random-init network overflows fp32 itself under that: undone in fc.3, h still > 65504)
        sd["conv_layers.0.fc.0.bias"] = sd["conv_layers.0.fc.0.bias"] * 3e5
        sd["conv_layers.0.fc.3.weight"] = sd["conv_layers.0.fc.3.weight"] / 3e5
    want = OracleScoreModel(case.oracle_config(), sd)(case.make_batch())
    assert all(torch.isfinite(w).all() for w in want)
    model = _model_for(case, sd)
    b = case.make_batch().to(_dev())
    got = model(b)
    assert model.__dict__.get("h2_recoveries", 0) == 1
    for g, w, k in zip(got, want, ("tr", "rot", "tor", "sc_tor")):
        if w.numel():
            assert rel_err(g.float().cpu(), w) < TOL, (k, rel_err(g.float().cpu(), w))
    model.check_overflow()                                                          # (nothing left behind for the next forward)
    got2 = model(b)                                                                 # ... which recovers again, same values
    assert model.__dict__["h2_recoveries"] == 2
    for g, g2 in zip(got, got2):
        assert torch.equal(g, g2)
    # a sampler's run: captured steps in the h2 form, one check at the end, the whole job again in the fp32 form - the poses are those of
    # a model that runs the fp32 form from the start
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    g3 = make_3dpf_complex(seed=0, flexible_sidechains=False)
    import bench
    cfgname = "cfg1" if name == "cfg1_full" else "cfg2"
    poses = {}
    for first_try_h2 in (True, False):
        m2, kw = bench.build_model(cfgname, False, _dev())
        with torch.no_grad():     # (h = relu(fc1) of conv 0 far beyond 65504, undone in fc.3: the job stays the seeded model's job)
            m2.conv_layers[0].fc[0].weight.mul_(3e5)
            m2.conv_layers[0].fc[0].bias.mul_(3e5)
            m2.conv_layers[0].fc[3].weight.mul_(1.0 / 3e5)
        m2.conv_h2 = first_try_h2
        smp = Sampler(m2, g3, 4, _dev(), SamplerConfig(inference_steps=6, flexible_sidechains=False), seed=0)
        smp.randomize()
        lig, _ = smp.run(get_t_schedule(6))
        poses[first_try_h2] = lig.clone()
        assert m2.__dict__.get("h2_recoveries", 0) == (1 if first_try_h2 else 0)
        assert m2.conv_h2 == first_try_h2                                            # (the switch is the caller's again)
        smp.close()
    assert torch.isfinite(poses[True]).all() and torch.equal(poses[True], poses[False])


def test_rows_kernel_range_flag_is_raised_and_recovered():
    """the recovery test above scales fc.0 by 3e5, which puts |w| beyond the row-stationary kernel's weight planes (255) and
    moves the whole model to the 32-edge kernel - ddp_conv_rows' own flag site never ran.  Here the weights stay inside the planes
    (launch.rows_mode stays on for every factorised conv) and a VALUE leaves the kernel's range: fc.0 (weight and bias) x 3000 and
    fc.3 / 3000 on one conv (and its edge embeddings x 100): |w| <= 3000 x 0.075 < 255, h = relu(fc1) of many edges beyond the h plane's 4094 (65504 / DDP_ROWS_SH) but
    far inside the 65504 of the 32-edge kernel's 2048-scaled planes.  The forward returns the fp32 form's scores (1e-4 of the oracle),
    counts one recovery, and the next forward does the same.  (Stage A's flag - a plane value |32 G| beyond the range - is raised at unit
    level: test_stage_a_plane_forms_report_values_outside_their_range.)"""
    from diffdock_pocket_amd import launch as K
    case, gold, batch, sd = case_inputs("cfg2_small")
    sd = dict(sd)
    conv = "conv_layers.12"       # (layer 1, atom<-atom: a factorised conv of the rows kernel)
    sd[conv + ".fc.0.weight"] = sd[conv + ".fc.0.weight"] * 3000.0
    sd[conv + ".fc.0.bias"] = sd[conv + ".fc.0.bias"] * 3000.0
    sd[conv + ".fc.3.weight"] = sd[conv + ".fc.3.weight"] / 3000.0
    # (... and the atom-atom edge embeddings x 100: a third of edge_attr_'s columns, so that h = relu(fc1) passes 4094 on many edges with room
    # to spare - the weights alone stop at 255 / 0.075 = 3400)
    sd["atom_edge_embedding.3.weight"] = sd["atom_edge_embedding.3.weight"] * 100.0
    sd["atom_edge_embedding.3.bias"] = sd["atom_edge_embedding.3.bias"] * 100.0
    assert float(sd[conv + ".fc.0.weight"].abs().max()) < 255.0 and float(sd[conv + ".fc.3.weight"].abs().max()) < 255.0
    want = OracleScoreModel(case.oracle_config(), sd)(case.make_batch())
    assert all(torch.isfinite(w).all() for w in want)
    model = _model_for(case, sd)
    dev = _dev()
    # (every factorised conv carries the row-stationary kernel's weight stream: none was pushed back to the 32-edge kernel by its weights)
    assert all(c.packed_g(dev).wsh is not None for c in model.conv_layers if getattr(c, "spec_g", None) is not None and c.spec_g.factorized)
    b = case.make_batch().to(dev)
    got = model(b)
    assert model.__dict__.get("h2_recoveries", 0) == 1
    for g, w, k in zip(got, want, ("tr", "rot", "tor", "sc_tor")):
        if w.numel():
            assert rel_err(g.float().cpu(), w) < TOL, (k, rel_err(g.float().cpu(), w))
    model.check_overflow()
    got2 = model(b)
    assert model.__dict__["h2_recoveries"] == 2
    for g, g2 in zip(got, got2):
        assert torch.equal(g, g2)


@pytest.mark.parametrize("flex", [False, True])
def test_pipelined_layer_order_is_bitwise_the_serial_one(flex):
    """For large batches the launches of a conv layer run as parallel branches of the captured step (engine._layers, "pipelined":
    the direct conv of layer l + 1 on a stream of its own from the moment the atom and receptor means of layer l are queued, the
    receptor / ligand / atom chains [mean -> stage A] side by side).  Same kernels, same per-edge arithmetic: the scores of forwards
    at two schedule positions, launch by launch and through a replayed hipGraph, and the poses must be bit for bit those of the serial
    order (model.overlap_direct_conv = False) - with a layer's factorised convs as two launches (model.split_rows_launch, the default) and as
    one.  Eight samples with the small-batch fork switched off (concurrent_max_atoms = 0: the
    large-batch path whatever the batch size) - a captured 40-sample step holds ~40 GB of device memory; the 40-sample batch runs
    this order in test_bench_batch_samples_match_oracle and in bench.py."""
    import bench
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    sched = get_t_schedule(20)
    g = make_3dpf_complex(seed=0, flexible_sidechains=flex)
    out = {}
    orders = ("serial", "pipelined", "pipelined, one conv launch per layer")
    for order in orders:
        model, kw = bench.build_model("cfg2", flex, dev)
        model.concurrent_max_atoms = 0
        model.overlap_direct_conv = order != "serial"
        assert model.split_rows_launch          # (the default: receptor- / ligand-sourced convs beside stage A of the atom rows)
        model.split_rows_launch = order == "pipelined"
        model.split_rows_min_g_bytes = 0       # (the default threshold, 4 GB of G for the atom rows, is the 40-sample batch)
        smp = Sampler(model, g, 8, dev, SamplerConfig(inference_steps=20, flexible_sidechains=flex), seed=0)
        smp.randomize()
        res = [[t.clone() for t in smp.scores(float(sched[0]))]]
        for i in range(4):       # two ordinary steps, the capture, one replay
            smp.step(i, sched)
        assert bool(smp._graph)
        res.append([t.clone() for t in smp.scores(float(sched[4]))])
        res.append([smp.lig_pos.clone(), smp.atom_pos.clone()])
        out[order] = res
        smp.close()
        del smp, model
    for order in orders[1:]:
        for a, b in zip(out["serial"], out[order]):
            for x, y in zip(a, b):
                assert torch.equal(x, y), order


@pytest.mark.parametrize("flex,large", [(False, True), (False, False), (True, True)])
def test_forked_front_is_bitwise_the_serial_one(flex, large):
    """Round 4: the front's independent chains run as parallel branches of the captured step (model.fork_front): [node encoders ->
    edge embeddings] beside [neighbour searches -> views], and with a rigid receptor the index lists of the work eliminations beside
    stage A + the conv launch of layer 0 (engine._forward / _front / _layers).  Same kernels, same arguments: scores at two schedule
    positions (launch by launch, and after a capture + replay) and the poses are bit for bit those of fork_front = False, on the
    large-batch launch order (concurrent_max_atoms = 0) and on the small-batch one."""
    import bench
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    sched = get_t_schedule(20)
    g = make_3dpf_complex(seed=0, flexible_sidechains=flex)
    out = {}
    for ff in (False, True):
        model, kw = bench.build_model("cfg2", flex, dev)
        if large:
            model.concurrent_max_atoms = 0
        model.fork_front = ff
        model.fork_small_means = ff      # (small batches: the three segmented means of a layer side by side)
        smp = Sampler(model, g, 6, dev, SamplerConfig(inference_steps=20, flexible_sidechains=flex), seed=0)
        smp.randomize()
        res = [[t.clone() for t in smp.scores(float(sched[0]))]]
        for i in range(5):       # two ordinary steps, the capture, two replays
            smp.step(i, sched)
        assert bool(smp._graph)
        res.append([t.clone() for t in smp.scores(float(sched[5]))])
        res.append([smp.lig_pos.clone(), smp.atom_pos.clone()])
        out[ff] = res
        smp.close()
        del smp, model
    for a, b in zip(out[False], out[True]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_captured_steps_share_one_memory_pool():
    """A captured step keeps its intermediates in the graph's memory pool; on this ROCm stack capture-time memory that PyTorch hands
    back to the driver does not come back (a pool per graph lost 35 - 45 GB per captured 40-sample sampler).  The captured steps of a
    process share ONE pinned pool per device (sampler.Sampler._graph_pool): a second sampler, captured after the first was closed,
    reuses its segments - the device's used memory does not grow by another graph's worth - and replays correctly."""
    import bench
    import gc
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    sched = get_t_schedule(20)
    g = make_3dpf_complex(seed=0, flexible_sidechains=False)
    model, kw = bench.build_model("cfg2", False, dev)

    def used():
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        return (total - free) / 1e9

    poses, after = [], []
    for rep in range(3):
        smp = Sampler(model, g, 8, dev, SamplerConfig(inference_steps=20, flexible_sidechains=False), seed=0)
        smp.randomize()
        for i in range(5):
            smp.step(i, sched)
        assert bool(smp._graph)
        poses.append(smp.lig_pos.clone())
        smp.close()
        del smp
        gc.collect()
        after.append(used())
    assert torch.equal(poses[0], poses[1]) and torch.equal(poses[0], poses[2])
    assert after[2] - after[0] < 1.0, after        # GB: no second / third graph's worth of memory (one 8-sample step is ~8 GB)


def test_a_truncated_ligand_atom_edge_list_is_reported():
    """The ligand<-atom edge list has a capacity per ligand atom (model.la_capacity_per_atom) instead of its worst case.  A
    search that finds more pairs drops them AND raises a flag in pinned host memory: the next forward refuses to go on."""
    from diffdock_pocket_amd import _lib as L
    case, gold, batch, sd = case_inputs("cfg2_small")
    model = _model_for(case, sd)
    b = case.make_batch().to(_dev())
    want = [t.clone() for t in model(b)]
    model.la_capacity_per_atom = 2
    model._static_cache = {}
    model(b)
    torch.cuda.synchronize()
    with pytest.raises(L.DdpError, match="la_capacity_per_atom"):
        model(b)
    model.la_capacity_per_atom = 128
    model._static_cache = {}
    for x, y in zip(model(b), want):        # (the flag was cleared by the report)
        assert torch.equal(x, y)


def test_device_driven_step_equals_the_host_driven_one():
    """tests/golden/step_outputs_r02_host_path.pt holds the scores and poses of the first denoising steps of bench.py's jobs as
    the round-2 forward produced them (exact-size lists, ~10 host synchronisations and ~700 PyTorch launches per step; written
    by tools/dump_step_outputs.py at commit 'Node encoders ... as one HIP launch').  Up to the commit that moved the read-out
    MLPs into ddp_trrot_head / ddp_tor_head the device-driven step reproduced them BIT FOR BIT (same kernels, same per-element
    arithmetic, same summation orders - only who knows the list sizes had changed; profiles/r03_v2_pytest_gpu.txt).  Since then
    the read-out MLPs sum their 33 / 120 products in index order instead of hipBLASLt's order, the ligand centre is summed in
    index order instead of torch.sum's and the node encoders add the step-independent part of their Linear as one term (stage A
    is the exact fp32 form: model.stage_a_bf16x3 stays False): the edge counts are still identical, scores and poses agree to
    2e-5 of the largest component (the parity tolerance of the path is 1e-4).  This is a REGRESSION guard against the product's
    own round-2 output, not parity: parity of whole trajectories against the oracle-driven CPU sampler is
    test_cfg1_job_end_to_end_against_the_cpu_sampler below."""
    STEP_TOL = 2e-5
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import dump_step_outputs as D
    ref = torch.load(os.path.join(root, "tests", "golden", "step_outputs_r02_host_path.pt"), weights_only=False)
    dev = _dev()
    for cfg, flex, n in D.JOBS:
        got = D.run_job(cfg, flex, n, dev)
        want = ref[f"{cfg}_flex{int(flex)}_n{n}"]
        for t_idx in D.STEPS:
            for k in ("E_ll", "E_lr", "E_la", "E_aa"):
                assert got[t_idx]["stats"][k] == want[t_idx]["stats"][k], (cfg, flex, n, t_idx, k)
            for name, a, b in zip(("tr", "rot", "tor", "sc_tor"), got[t_idx]["scores"], want[t_idx]["scores"]):
                assert elementwise_excess(a, b, STEP_TOL, STEP_TOL) <= 1, (cfg, flex, n, t_idx, name, rel_err(a, b))
            # (coordinates are ~10 - 50 A: a few ulp of the largest one)
            assert elementwise_excess(got[t_idx]["lig_pos"], want[t_idx]["lig_pos"], STEP_TOL, STEP_TOL) <= 1, (cfg, flex, n, t_idx)
            if "atom_pos" in want[t_idx]:
                assert elementwise_excess(got[t_idx]["atom_pos"], want[t_idx]["atom_pos"], STEP_TOL, STEP_TOL) <= 1, (cfg, flex, n, t_idx)


def test_weight_edits_through_param_data_invalidate_the_packed_weights():
    """The reference's EMA writes weights with `param.data.copy_` (utils/utils.py:216,239), which no autograd version counter
    sees.  After a first forward (packed weights, stage-A stacks and edge packs built) an `fc.3.weight`, an edge-embedding
    weight and a BatchNorm buffer are overwritten through `.data`: the next forward must equal, bit for bit, a fresh model
    loaded with the edited weights - and an unchanged model must keep its packed weights (no repacking per call)."""
    case, gold, batch, sd = case_inputs("cfg2_small")
    dev = _dev()
    model = _model_for(case, sd)
    b = case.make_batch().to(dev)
    out0 = [t.clone() for t in model(b)]
    packed_before = model.conv_layers[0]._packed
    out0b = [t.clone() for t in model(b)]
    assert model.conv_layers[0]._packed is packed_before            # nothing changed: caches kept
    assert all(torch.equal(x, y) for x, y in zip(out0, out0b))
    g = torch.Generator().manual_seed(5)
    sd2 = {k: v.clone() for k, v in sd.items()}
    for key in ("conv_layers.4.fc.3.weight", "lig_edge_embedding.0.weight", "conv_layers.1.batch_norm.running_var"):
        sd2[key] = sd2[key] * (1.0 + 0.2 * torch.rand(sd2[key].shape, generator=g))
    params = dict(model.named_parameters())
    params.update(dict(model.named_buffers()))
    for key in ("conv_layers.4.fc.3.weight", "lig_edge_embedding.0.weight", "conv_layers.1.batch_norm.running_var"):
        v0 = params[key]._version
        params[key].data.copy_(sd2[key].to(dev))
        assert params[key]._version == v0                           # invisible to the version counters
    out1 = [t.clone() for t in model(b)]
    fresh = _model_for(case, sd2)
    want = fresh(case.make_batch().to(dev))
    assert all(torch.equal(x, y) for x, y in zip(out1, want))
    assert not all(torch.equal(x, y) for x, y in zip(out1, out0))


@pytest.mark.parametrize("name", ["cfg2_small", "cfg1_edge"])
def test_forward_direct_path_matches_too(name):
    """factorize_min_degree = 0 forces every conv onto the direct (per-edge MFMA) path; both paths are kept correct."""
    case, gold, batch, sd = case_inputs(name)
    model = _model_for(case, sd)
    model.factorize_min_degree = 0
    got = model(batch.to(_dev()))
    for g, k in zip(got, ("tr", "rot", "tor", "sc_tor")):
        assert rel_err(g.float().cpu(), gold["outputs"][k]) < TOL, (name, k)


@pytest.mark.parametrize("name", ["cfg2_small", "cfg2_full_noflex"])
def test_direct_convs_as_tasks_of_segment_ranges_change_no_bit(name):
    """model.direct_rows: the layers' direct convs through ddp_conv_rows16_direct_kernel; model.direct_rows_max_split = n: each of them as up
    to n tasks of output-segment ranges (ddp_conv_task_t::rows_seg0 / rows_seg1 / rows_nts, a weight stream per range).  Every output column
    is computed by exactly one task from the same tiles in the same order, so the scores of n = 2, 3, 6 are those of n = 1 BIT FOR BIT; and
    against ddp_conv_messages (direct_rows off: 2048-scaled planes, 64-edge workgroups) within the path's tolerance of the golden outputs."""
    case, gold, batch, sd = case_inputs(name)
    dev = _dev()
    outs = {}
    for n in (1, 2, 3, 6):
        model = _model_for(case, sd)
        assert model.direct_rows and model.rows_mfma16
        model.direct_rows_max_split = n
        outs[n] = [t.clone() for t in model(case.make_batch().to(dev))]
        for g, k in zip(outs[n], ("tr", "rot", "tor", "sc_tor")):
            if gold["outputs"][k].numel():
                assert rel_err(g.float().cpu(), gold["outputs"][k]) < TOL, (name, n, k)
    for n in (2, 3, 6):
        assert all(torch.equal(a, b) for a, b in zip(outs[1], outs[n])), (name, n)
    model = _model_for(case, sd)
    model.direct_rows = False
    off = model(case.make_batch().to(dev))
    for g, k in zip(off, ("tr", "rot", "tor", "sc_tor")):
        if gold["outputs"][k].numel():
            assert rel_err(g.float().cpu(), gold["outputs"][k]) < TOL, (name, "direct_rows off", k)


def test_forward_is_deterministic():
    case, gold, batch, sd = case_inputs("cfg1_full")
    model = _model_for(case, sd)
    a = [t.clone() for t in model(case.make_batch().to(_dev()))]
    b = model(case.make_batch().to(_dev()))
    for x, y in zip(a, b):
        assert torch.equal(x, y)     # ordered reductions: bitwise reproducible


@pytest.mark.parametrize("ns,nv,layer,E,N", [(16, 4, 0, 1, 3), (16, 4, 1, 63, 10), (16, 4, 2, 64, 10), (24, 6, 3, 65, 7),
                                              (60, 10, 3, 200, 23), (60, 10, 0, 129, 5), (32, 6, 3, 500, 40),
                                              (64, 32, 3, 70, 9),
                                              # vector blocks wider than one 16-column tile of the row-stationary kernel (nv > 16): two column tiles in
                                              # their G runs, layer 0 without and layer 1 with stream tiles (24 features: the general feature path)
                                              (60, 20, 0, 150, 9), (32, 24, 1, 170, 11)])
@pytest.mark.parametrize("factorized,form,fmt", [(False, 0, 0), (False, 1, 0), (True, 1, 1), (True, 1, 0), (True, 0, 1), (True, 0, 0)])
def test_single_conv_layer(ns, nv, layer, E, N, factorized, form, fmt):
    """TensorProductConvLayer.forward with the reference call signature (models/score_model.py:108) on ragged
    edge sets: partial tiles, receivers without edges, repeated receivers.  Factorised: through both forms of the row-stationary kernel
    (ddp_conv_task_t::rows_form: 1 = v_mfma_f32_16x16x32_f16, the model's default) and both plane forms of G (gh_fmt: 1 = fp16 hi +
    continuation byte, the model's default) where the shape is one of that kernel's - a stand-alone layer has no model to set them; direct
    with form 1: through ddp_conv_rows16_direct_kernel (model.direct_rows: every feature a stream tile, the bias in k, feature chunks)."""
    from diffdock_pocket_amd import packing as P
    from diffdock_pocket_amd.score_model import TensorProductConvLayer
    torch.manual_seed(ns + layer + E)
    mi, mo = P.irreps_muls(ns, nv, layer), P.irreps_muls(ns, nv, layer + 1)
    spec = P.faster_tp_spec(mi, mo, 3 * ns)
    blocks = [(m, d, s) for m, d, s in ((mo[0], 1, True), (mo[1], 3, False), (mo[2], 3, False), (mo[3], 1, False)) if m]
    conv = TensorProductConvLayer(spec, blocks, spec_g=P.faster_tp_spec(mi, mo, 3 * ns, factorized=True))
    with torch.no_grad():
        conv.batch_norm.running_mean.normal_(0, 0.2)
        conv.batch_norm.running_var.uniform_(0.5, 2)
        conv.batch_norm.weight.uniform_(0.5, 1.5)
        conv.batch_norm.bias.normal_(0, 0.2)
    x = torch.randn(N, P.irreps_dim(mi))
    ei = torch.stack([torch.randint(0, max(N - 1, 1), (E,)), torch.randint(0, N, (E,))])   # node N-1 never receives
    ea = torch.randn(E, 3 * ns)
    sh = tp.spherical_harmonics("1x0e+1x1o", torch.randn(E, 3))
    cfg = OracleConfig(ns=ns, nv=nv)
    sd = {"c." + k: v for k, v in conv.state_dict().items()}
    want = OracleScoreModel(cfg, sd)._conv("c", cfg.irreps(layer), cfg.irreps(layer + 1), x, ei, ea, sh)
    dev = _dev()
    conv = conv.to(dev)
    conv.rows_form, conv.gh_fmt = form, fmt
    conv.direct_rows = bool(form) and not factorized      # (False, 1, 0): the DIRECT conv through ddp_conv_rows16_direct_kernel where the shape allows it (ns = 60)
    got = conv(x.to(dev), ei.to(dev), ea.to(dev), sh.to(dev), factorized=factorized).cpu()
    if conv.direct_rows and ns == 60:
        assert conv.packed_rows_direct(dev) is not None and conv.packed_rows_direct(dev).rows_bias_k == 1
    if factorized and ns in (60, 32) and P.rows_supported(conv.spec_g):
        pk = conv.packed_g(dev)
        assert pk.wsh is not None and pk.rows_form == form and pk.gh_fmt == fmt      # (it ran through the kernel form and plane form asked for)
    assert got.shape == want.shape
    assert rel_err(got, want) < 2e-5, rel_err(got, want)
    # empty edge set: scalar zero, like the reference (models/score_model.py:109-111)
    z = conv(x.to(dev), ei[:, :0].to(dev), ea[:0].to(dev), sh[:0].to(dev))
    assert z.dim() == 0 and float(z) == 0.0


@pytest.mark.parametrize("form,fmt", [(1, 1), (0, 0)])
@pytest.mark.parametrize("mag", [1.0, 1e-2, 1e-4])
def test_rows_kernel_operand_planes_at_small_magnitudes(mag, form, fmt):
    """ddp_conv_rows computes on UNIFIED fp16 hi/lo planes (V = v 2^s = hi + lo, both halves at one scale: include/ddp_hip.h DDP_ROWS_S*):
    22 significant bits while lo is a normal fp16 number, an ABSOLUTE floor of 2^-25 / 2^s per operand element below that (2^-29 for
    edge_attr_ and h, 2^-33 for the fc weights, 2^-30 for G).  Asserted here on a factorised ns = 60 conv whose edge_attr_ AND node
    features are scaled down to 1e-2 and 1e-4 (the model's own operands are O(1e-2 ... 10)): the messages stay within
    2e-5 max|message| + the floor's share, K 2^-29 max|w| per product, of the oracle's."""
    from diffdock_pocket_amd import launch as K
    from diffdock_pocket_amd import packing as P
    from diffdock_pocket_amd.score_model import TensorProductConvLayer
    ns, nv, layer, E, N = 60, 10, 3, 700, 60
    torch.manual_seed(11)
    mi, mo = P.irreps_muls(ns, nv, layer), P.irreps_muls(ns, nv, layer + 1)
    spec = P.faster_tp_spec(mi, mo, 3 * ns)
    blocks = [(m, d, s) for m, d, s in ((mo[0], 1, True), (mo[1], 3, False), (mo[2], 3, False), (mo[3], 1, False)) if m]
    conv = TensorProductConvLayer(spec, blocks, batch_norm=False, spec_g=P.faster_tp_spec(mi, mo, 3 * ns, factorized=True))
    x = torch.randn(N, P.irreps_dim(mi)) * mag
    src = torch.sort(torch.randint(0, N, (E,)))[0]
    ei = torch.stack([torch.randint(0, N, (E,)), src])
    ea = torch.randn(E, 3 * ns) * mag
    sh = tp.spherical_harmonics("1x0e+1x1o", torch.randn(E, 3))
    cfg = OracleConfig(ns=ns, nv=nv, batch_norm=False)
    sd = {"c." + k: v for k, v in conv.state_dict().items()}
    want = OracleScoreModel(cfg, sd)._conv("c", cfg.irreps(layer), cfg.irreps(layer + 1), x, ei, ea, sh)
    dev = _dev()
    conv = conv.to(dev)
    conv.rows_form, conv.gh_fmt = form, fmt      # (1, 1: the model's defaults - the 16x16x32 kernel, G as fp16 hi + continuation byte: absolute 2^-24 / 32 on G below |32 G| = 2^-14)
    assert K.CONV_ROWS and conv.packed_g(dev).wsh is not None      # (the row-stationary kernel's weight stream exists: this conv runs through it)
    got = conv(x.to(dev), ei.to(dev), ea.to(dev), sh.to(dev), factorized=True).cpu()
    rows_was = K.CONV_ROWS
    K.CONV_ROWS = False
    try:
        ref32 = conv(x.to(dev), ei.to(dev), ea.to(dev), sh.to(dev), factorized=True).cpu()     # the 32-edge kernel's 2048-scaled planes
    finally:
        K.CONV_ROWS = rows_was
    base = want
    wmax = float(max(conv.fc[0].weight.abs().max(), conv.fc[3].weight.abs().max()))
    floor = 192 * 2.0 ** -29 * wmax * 4.0      # K products per output, a few outputs summed per message element
    tol = 2e-5 * float(base.abs().max()) + floor
    assert float((got - base).abs().max()) <= tol, (mag, float((got - base).abs().max()), tol)
    assert float((got - ref32).abs().max()) <= tol, (mag, float((got - ref32).abs().max()), tol)


def test_edge_featurize_and_torsion_sh():
    import ctypes as C
    from diffdock_pocket_amd import _lib as L
    from diffdock_pocket_amd.score_model import GaussianSmearing, _EdgeMLPPack, _edge_featurize, _ptr, _stream
    torch.manual_seed(0)
    dev = _dev()
    ns, k, E, Na, Nb = 60, 64, 1000, 50, 70
    seq = torch.nn.Sequential(torch.nn.Linear(10 + k, ns), torch.nn.ReLU(), torch.nn.Dropout(0.0), torch.nn.Linear(ns, ns))
    dist = GaussianSmearing(0.0, 5.0, k)
    pa, pb = torch.randn(Na, 3) * 3, torch.randn(Nb, 3) * 3
    ia, ib = torch.randint(0, Na, (E,)), torch.randint(0, Nb, (E,))
    pb[ib[0]] = pa[ia[0]]                                   # a zero-length edge: sh = [1,0,0,0]
    other = torch.randn(E, 10)
    vec = pb[ib] - pa[ia]
    want = seq(torch.cat([other, gaussian_smearing(vec.norm(dim=-1), dist.offset)], 1)).detach()
    want_sh = tp.spherical_harmonics("1x0e+1x1o", vec)
    seq, dist = seq.to(dev), dist.to(dev)
    pk = _EdgeMLPPack(seq, slice(10, 10 + k), dev)
    pre = other.to(dev) @ pk.W1[:, :10].t() + pk.b1
    out, sh = _edge_featurize(pk, dist, pa.to(dev), ia.int().to(dev), pb.to(dev), ib.int().to(dev), pre,
                              torch.arange(E, dtype=torch.int32, device=dev))
    assert rel_err(out.cpu(), want) < 1e-5
    assert float((sh.cpu() - want_sh).abs().max()) < 1e-5
    # torsion harmonics vs restated FullTensorProduct
    T = 9
    bv = torch.randn(T, 3)
    boe = torch.randint(0, T, (E,))
    y2 = tp.spherical_harmonics("2e", bv)
    want_t = tp.FullTensorProduct("1x0e+1x1o", "2e")(want_sh, y2[boe])[:, :3]
    got_t = torch.empty(E, 4, device=dev)
    bv_d, boe_d = bv.to(dev).contiguous(), boe.int().to(dev)   # keep the device buffers alive across the launch
    # ... and, in the same launch, the bonds' node attributes x[b0, :ns] + x[b1, :ns] (all_atom_score_model.py:399,423)
    x = torch.randn(Nb, 184, device=dev)
    b0, b1 = torch.randint(0, Nb, (T,), device=dev).int(), torch.randint(0, Nb, (T,), device=dev).int()
    battr = torch.empty(T, ns, device=dev)
    L.check(L.load().ddp_torsion_sh(_ptr(sh), _ptr(bv_d), _ptr(boe_d), E, None, _ptr(got_t), _ptr(x), 184, ns, _ptr(b0), _ptr(b1), T,
                                    _ptr(battr), _stream()), "ddp_torsion_sh")
    assert float((got_t.cpu()[:, 1:] - want_t).abs().max()) < 1e-5 and float(got_t[:, 0].abs().max()) == 0.0
    assert torch.equal(battr, x[b0.long(), :ns] + x[b1.long(), :ns])


def test_edge_featurize_jobs_equal_single_launches():
    """ddp_edge_featurize_jobs: several edge sets in one launch - different MLPs, sizes from 1 edge to several workgroups, a
    device-side count below the capacity, an empty set - bitwise the single-set launches."""
    from diffdock_pocket_amd import launch as K
    from diffdock_pocket_amd.score_model import GaussianSmearing, _EdgeMLPPack
    torch.manual_seed(1)
    dev = _dev()
    calls = []
    for ns, k, E, n_dev in ((60, 32, 5000, None), (60, 64, 1, None), (16, 32, 700, 333), (24, 32, 0, None), (60, 32, 129, 129)):
        seq = torch.nn.Sequential(torch.nn.Linear(7 + k, ns), torch.nn.ReLU(), torch.nn.Dropout(0.0), torch.nn.Linear(ns, ns)).to(dev)
        dist = GaussianSmearing(0.0, 5.0, k).to(dev)
        Na, Nb = 50, 70
        pa, pb = torch.randn(Na, 3, device=dev) * 3, torch.randn(Nb, 3, device=dev) * 3
        ia, ib = torch.randint(0, Na, (E,), device=dev).int(), torch.randint(0, Nb, (E,), device=dev).int()
        pk = _EdgeMLPPack(seq, slice(7, 7 + k), dev)
        pre = torch.randn(Na, ns, device=dev)
        kw = {}
        if n_dev is not None:
            kw = dict(n_edges=E, cnt=torch.tensor([n_dev], dtype=torch.int32, device=dev))
        calls.append(((pk, dist, pa, ia, pb, ib, pre, ia), kw))
    got = K.edge_featurize_jobs(calls)
    for (a, kw), (o, s_) in zip(calls, got):
        wo, ws = K.edge_featurize(*a, **kw)
        n = int(kw["cnt"][0]) if kw else o.shape[0]
        assert o.shape == wo.shape and torch.equal(o[:n], wo[:n]) and torch.equal(s_[:n], ws[:n])


@pytest.mark.parametrize("B,stride0", [(1, False), (40, True), (130, False)])
def test_prologue_and_read_out_kernels_match_their_pytorch_definitions(B, stride0):
    """ddp_step_prologue / ddp_trrot_head / ddp_tor_head (csrc/ddp_heads.hip) against the PyTorch expressions they replace
    (engine.py of round 2 = all_atom_score_model.py:244-245,362-384,400-410,548-550,571-576,589-592)."""
    import ctypes as C
    import functools
    from diffdock_pocket_amd import _lib as L
    from diffdock_pocket_amd import launch as K
    from diffdock_pocket_amd.diffusion import SigmaRanges, _frequencies, sinusoidal_embedding, t_to_sigma
    from diffdock_pocket_amd.score_model import TensorProductScoreModel
    dev = _dev()
    lib = L.load()
    torch.manual_seed(B)
    kw = dict(CASES["cfg1_full"].model_kwargs())
    kw.update(CASES["cfg1_full"].ctor_extras())
    kw["device"] = dev
    m = TensorProductScoreModel(**kw).to(dev).eval()
    ns, sd = m.ns, m.sigma_embed_dim
    rng = SigmaRanges()
    assert m._sigma_ranges() is not None or not isinstance(m.t_to_sigma, functools.partial)
    # ---- prologue
    if stride0:
        tbuf = torch.rand(4, device=dev)
        ts = [tbuf[k:k + 1].expand(B) for k in range(4)]
    else:
        ts = [torch.rand(B, device=dev) for _ in range(4)]
    sizes = torch.randint(1, 40, (B,))
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).int().to(dev)
    Nl = int(ptr[-1])
    pos = torch.randn(Nl, 3, device=dev) * 5
    T = 3 * B
    b0, b1 = torch.randint(0, Nl, (T,), device=dev).int(), torch.randint(0, Nl, (T,), device=dev).int()
    src = torch.randint(0, 1000, (77,), device=dev).int()
    a = L.PrologueArgs()
    sig, cut, emb, cen = torch.empty(4, B, device=dev), torch.empty(B, device=dev), torch.empty(B, sd, device=dev), torch.empty(B, 3, device=dev)
    mid, vec, dst = torch.empty(T, 3, device=dev), torch.empty(T, 3, device=dev), torch.zeros(80, dtype=torch.int32, device=dev)
    freq = _frequencies(sd // 2, 10000, dev)
    scale = m.timestep_emb_func.keywords.get("scale", 1.0)
    a.n_graphs = B
    lohi = [(rng.tr_sigma_min, rng.tr_sigma_max), (rng.rot_sigma_min, rng.rot_sigma_max), (rng.tor_sigma_min, rng.tor_sigma_max),
            (rng.sidechain_tor_sigma_min, rng.sidechain_tor_sigma_max)]
    for k in range(4):
        a.t[k], a.t_stride[k], a.sigma[k] = ts[k].data_ptr(), ts[k].stride(0) if B > 1 else 0, sig[k].data_ptr()
        a.sig_min[k], a.sig_max[k] = lohi[k]
    a.cut, a.cut_mul, a.cut_add = cut.data_ptr(), 3.0, 20.0
    a.graph_emb, a.sd, a.emb_scale, a.freq = emb.data_ptr(), sd, scale, freq.data_ptr()
    a.lig_pos, a.graph_ptr, a.center = pos.data_ptr(), ptr.data_ptr(), cen.data_ptr()
    a.bonds[1].pos, a.bonds[1].b0, a.bonds[1].b1, a.bonds[1].n = pos.data_ptr(), b0.data_ptr(), b1.data_ptr(), T
    a.bonds[1].mid, a.bonds[1].vec = mid.data_ptr(), vec.data_ptr()
    a.copy[0].src, a.copy[0].dst, a.copy[0].n = src.data_ptr(), dst.data_ptr(), 77
    L.check(lib.ddp_step_prologue(C.byref(a), K.stream()), "ddp_step_prologue")
    want_sig = t_to_sigma(*[t.contiguous() for t in ts], args=rng)
    for k in range(4):
        assert rel_err(sig[k], want_sig[k]) < 1e-6, k
    assert rel_err(cut, want_sig[0] * 3 + 20) < 1e-6
    assert float((emb.cpu() - sinusoidal_embedding(ts[0].contiguous().cpu(), sd, scale=scale)).abs().max()) < 3e-5
    want_cen = torch.stack([pos[int(ptr[g]):int(ptr[g + 1])].double().mean(0) for g in range(B)]).float()
    assert float((cen - want_cen).abs().max()) < 1e-5
    assert torch.equal(mid, (pos[b0.long()] + pos[b1.long()]) / 2) and torch.equal(vec, pos[b1.long()] - pos[b0.long()])
    assert torch.equal(dst[:77], src) and int(dst[77:].abs().sum()) == 0
    # sigma as an INPUT (a foreign t_to_sigma): the cutoff follows it
    a.sig_max[0] = 0.0
    sig[0].fill_(2.0)
    L.check(lib.ddp_step_prologue(C.byref(a), K.stream()), "ddp_step_prologue")
    assert torch.equal(cut, torch.full_like(cut, 26.0)) and torch.equal(sig[0], torch.full_like(cut, 2.0))
    # ---- tr / rot read-out
    hw = m._head_weights(dev)
    gp = torch.randn(B, 12, device=dev)
    tr_sigma, rot_sigma = want_sig[0].contiguous(), want_sig[1].contiguous()
    emb_t = sinusoidal_embedding(ts[0].contiguous(), sd, scale=scale)
    r = L.TrRotArgs()
    out_tr, out_rot = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
    r.gp, r.ld_gp, r.n_graphs, r.ns, r.sd, r.graph_emb = gp.data_ptr(), 12, B, ns, sd, emb_t.data_ptr()
    for i, name in enumerate(("tr_final_layer", "rot_final_layer")):
        r.w1[i], r.b1[i], r.w2[i], r.b2[i] = (t.data_ptr() for t in hw[name])
    r.sigma[0], r.sigma[1] = tr_sigma.data_ptr(), rot_sigma.data_ptr()
    r.so3_table, r.so3_n, r.so3_lo, r.so3_span = hw["so3"].data_ptr(), hw["so3"].shape[0], hw["so3_lo"], hw["so3_span"]
    r.out[0], r.out[1] = out_tr.data_ptr(), out_rot.data_ptr()
    L.check(lib.ddp_trrot_head(C.byref(r), K.stream()), "ddp_trrot_head")
    with torch.no_grad():
        tr = gp[:, :3] + gp[:, 6:9]
        rot = gp[:, 3:6] + gp[:, 9:]
        n_tr, n_rot = tr.norm(dim=1, keepdim=True), rot.norm(dim=1, keepdim=True)
        want_tr = tr / n_tr * m.tr_final_layer(torch.cat([n_tr, emb_t], 1)) / tr_sigma.unsqueeze(1)
        want_rot = rot / n_rot * m.rot_final_layer(torch.cat([n_rot, emb_t], 1)) * m._so3_score_norm(rot_sigma).unsqueeze(1)
    assert elementwise_excess(out_tr, want_tr, 1e-5) <= 1 and elementwise_excess(out_rot, want_rot, 1e-5) <= 1
    # ---- torsion read-out
    h = torch.randn(T, 2 * ns, device=dev)
    gob = torch.randint(0, B, (T,), device=dev).int()
    tor_sigma = want_sig[2].contiguous()
    q = L.TorArgs()
    out = torch.empty(T, device=dev)
    q.h, q.ld_h, q.n_bonds, q.ns = h.data_ptr(), 2 * ns, T, ns
    q.w1, q.w2 = (t.data_ptr() for t in hw["tor_final_layer"])
    q.sigma, q.graph_of_bond = tor_sigma.data_ptr(), gob.data_ptr()
    q.torus_table, q.torus_n, q.torus_lo, q.torus_span = hw["torus"].data_ptr(), hw["torus"].shape[0] - 1, hw["torus_lo"], hw["torus_span"]
    q.out = out.data_ptr()
    L.check(lib.ddp_tor_head(C.byref(q), K.stream()), "ddp_tor_head")
    with torch.no_grad():
        want = m.tor_final_layer(h).squeeze(1) * torch.sqrt(m._torus_score_norm(tor_sigma[gob.long()]))
    assert elementwise_excess(out, want, 1e-5) <= 1


@pytest.mark.parametrize("k,ncols,nrows,ldx,offs", [(60, 12600, 300, 180, (120, 0)), (60, 70, 129, 180, (0,)),
                                                    (16, 980, 77, 37, (5, 3, 0)), (10, 33, 5, 12, (2,)), (64, 513, 200, 64, (0,))])
def test_stage_a_gemm(k, ncols, nrows, ldx, offs):
    """ddp_stage_a (weight-stationary VALU GEMM of the source-node factorisation, strided batch over (conv, slot)) against
    fp64 matmul; tolerance = fp32 rounding of a K-term sum."""
    import ctypes as C
    from diffdock_pocket_amd import _lib as L
    from diffdock_pocket_amd.score_model import _stream
    torch.manual_seed(k + ncols)
    dev = _dev()
    lib = L.load()
    nb = len(offs)
    x, w = torch.randn(nrows, ldx), torch.randn(nb, k, ncols)
    xd, wd = x.to(dev), w.to(dev)
    ldo = (ncols + 31) // 32 * 32 if ncols > 100 else ncols        # padded rows (the G layout) and dense rows (Gb)
    od = torch.full((nb, nrows, ldo), float("nan"), device=dev)
    L.check(lib.ddp_stage_a(xd.data_ptr(), ldx, nrows, None, None, nrows, (C.c_int32 * nb)(*offs), nb, wd.data_ptr(), None, k, ncols,
                            od.data_ptr(), ldo, _stream()), "ddp_stage_a")
    torch.cuda.synchronize()
    assert torch.isnan(od[:, :, ncols:]).all()                     # the padding columns are not written
    got = od[:, :, :ncols].cpu().double()
    assert torch.isfinite(got).all()
    for b in range(nb):
        want = x[:, offs[b]:offs[b] + k].double() @ w[b].double()
        assert float((got[b] - want).abs().max()) < 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("k,ncols,nrows,nb", [(60, 12672, 1111, 2), (60, 2560, 37, 3), (32, 3104, 500, 2), (60, 12672, 4500, 1)])
def test_stage_a_bf16x3_error(k, ncols, nrows, nb):
    """The bf16x3 form of stage A (both operands as three bfloat16 terms, six cross products accumulated in fp32 on
    v_mfma_f32_32x32x16_bf16) against an fp64 product: the error of an element stays below 2^-21 of sum_u |x w| - the class of an
    fp32 dot product of this length (the exact-fp32 MFMA form is measured beside it) - on rows with a device-side list, in place."""
    import ctypes as C
    from diffdock_pocket_amd import _lib as L
    from diffdock_pocket_amd.packing import split_bf16x3
    from diffdock_pocket_amd.score_model import _stream
    torch.manual_seed(k + nrows)
    dev = _dev()
    lib = L.load()
    ldx, ldo = 184, ncols
    x = (torch.randn(nrows, ldx) * torch.exp(torch.randn(nrows, 1))).to(dev)       # rows of very different magnitude
    w = (torch.randn(nb, k, ncols) * 0.1 * torch.exp(2 * torch.randn(nb, 1, ncols))).to(dev)
    offs = [0, 120, 60][:nb]
    offs_c = (C.c_int32 * nb)(*offs)
    w3 = split_bf16x3(w)
    assert w3.dtype == torch.bfloat16 and w3.shape == (nb, 3, (k + 15) // 16, 2, ncols, 8)
    exact = torch.stack([x[:, o:o + k].double() @ w[b].double() for b, o in enumerate(offs)])
    scale = torch.stack([x[:, o:o + k].double().abs() @ w[b].double().abs() for b, o in enumerate(offs)])
    out32, out3 = torch.empty(nb, nrows, ldo, device=dev), torch.full((nb, nrows, ldo), float("nan"), device=dev)
    L.check(lib.ddp_stage_a(x.data_ptr(), ldx, nrows, None, None, nrows, offs_c, nb, w.data_ptr(), None, k, ncols, out32.data_ptr(), ldo,
                            _stream()), "ddp_stage_a")
    L.check(lib.ddp_stage_a(x.data_ptr(), ldx, nrows, None, None, nrows, offs_c, nb, w.data_ptr(), w3.data_ptr(), k, ncols, out3.data_ptr(),
                            ldo, _stream()), "ddp_stage_a")
    e32 = float(((out32.double() - exact).abs() / scale).max())
    e3 = float(((out3.double() - exact).abs() / scale).max())
    assert e32 < 2.0 ** -21 and e3 < 2.0 ** -21, (e32, e3)
    # a row list with a device-side length: the listed rows only, bitwise the dense launch's rows
    rows = torch.randperm(nrows, device=dev)[:max(nrows // 3, 1)].int().contiguous()
    n_dev = torch.tensor([rows.numel() - 1], dtype=torch.int32, device=dev)
    part = torch.full((nb, nrows, ldo), -7.0, device=dev)
    L.check(lib.ddp_stage_a(x.data_ptr(), ldx, rows.numel(), rows.data_ptr(), n_dev.data_ptr(), nrows, offs_c, nb, w.data_ptr(), w3.data_ptr(),
                            k, ncols, part.data_ptr(), ldo, _stream()), "ddp_stage_a")
    sel = rows[:-1].long()
    assert torch.equal(part[:, sel], out3[:, sel])
    rest = torch.ones(nrows, dtype=torch.bool, device=dev)
    rest[sel] = False
    assert bool((part[:, rest] == -7.0).all())


@pytest.mark.parametrize("k,ncols,nrows,nb", [(60, 12672, 1111, 2), (60, 2560, 37, 3), (32, 3104, 500, 2), (24, 1856, 300, 2), (16, 1056, 77, 1),
                                              (60, 12672, 4500, 1)])
def test_stage_a_h2_error(k, ncols, nrows, nb):
    """The fp16 hi/lo split form of stage A (ddp_stage_a_h2: both operands as hi + lo / 2048, three products per 16 k on
    v_mfma_f32_32x32x16_f16, fp32 accumulation) against an fp64 product.  The operands carry 22 significant bits each, so a single
    product is off by at most 2^-21 + 2^-22 of |x w| (both roundings + the dropped xl wl); over all elements of these products - rows
    and columns of very different magnitude, dot products in which one term dominates - the worst element stays below 2^-20 of
    sum_u |x w| (measured 5.6e-7 where the exact fp32 MFMA chain, asserted beside it at 2^-21, shows 3.9e-7).  Also on a row list
    with a device-side length (the listed rows only, bitwise the dense launch's rows)."""
    import ctypes as C
    from diffdock_pocket_amd import _lib as L
    from diffdock_pocket_amd.packing import split_h2
    from diffdock_pocket_amd.score_model import _stream
    torch.manual_seed(k + nrows)
    dev = _dev()
    lib = L.load()
    ldx, ldo = 184, ncols
    x = (torch.randn(nrows, ldx) * torch.exp(torch.randn(nrows, 1))).to(dev)       # rows of very different magnitude
    w = (torch.randn(nb, k, ncols) * 0.1 * torch.exp(2 * torch.randn(nb, 1, ncols))).to(dev)
    offs = [0, 120, 60][:nb]
    offs_c = (C.c_int32 * nb)(*offs)
    wh = split_h2(w)
    assert wh.dtype == torch.float16 and wh.shape == (nb, 2, (k + 15) // 16, 2, ncols, 8)
    exact = torch.stack([x[:, o:o + k].double() @ w[b].double() for b, o in enumerate(offs)])
    scale = torch.stack([x[:, o:o + k].double().abs() @ w[b].double().abs() for b, o in enumerate(offs)])
    out32, outh = torch.empty(nb, nrows, ldo, device=dev), torch.full((nb, nrows, ldo), float("nan"), device=dev)
    L.check(lib.ddp_stage_a(x.data_ptr(), ldx, nrows, None, None, nrows, offs_c, nb, w.data_ptr(), None, k, ncols, out32.data_ptr(), ldo,
                            _stream()), "ddp_stage_a")
    L.check(lib.ddp_stage_a_h2(x.data_ptr(), ldx, nrows, None, None, nrows, offs_c, nb, w.data_ptr(), wh.data_ptr(), k, ncols, outh.data_ptr(),
                               ldo, None, _stream()), "ddp_stage_a_h2")
    assert not torch.equal(out32, outh)                                              # (another kernel did run)
    e32 = float(((out32.double() - exact).abs() / scale).max())
    eh = float(((outh.double() - exact).abs() / scale).max())
    assert e32 < 2.0 ** -21 and eh < 2.0 ** -20, (e32, eh)
    rows = torch.randperm(nrows, device=dev)[:max(nrows // 3, 1)].int().contiguous()
    n_dev = torch.tensor([rows.numel() - 1], dtype=torch.int32, device=dev)
    part = torch.full((nb, nrows, ldo), -7.0, device=dev)
    L.check(lib.ddp_stage_a_h2(x.data_ptr(), ldx, rows.numel(), rows.data_ptr(), n_dev.data_ptr(), nrows, offs_c, nb, w.data_ptr(), wh.data_ptr(),
                               k, ncols, part.data_ptr(), ldo, None, _stream()), "ddp_stage_a_h2")
    sel = rows[:-1].long()
    assert torch.equal(part[:, sel], outh[:, sel])
    rest = torch.ones(nrows, dtype=torch.bool, device=dev)
    rest[sel] = False
    assert bool((part[:, rest] == -7.0).all())


@pytest.mark.parametrize("mag", [1e-3, 1e-5, 3e-7])
def test_stage_a_h2_error_floor_for_small_operands(mag):
    """the split v = hi + lo / 2048 keeps 22 significant bits only while both halves are NORMAL fp16 numbers.  Below
    |v| ~ 6.1e-5 the hi half is subnormal (spacing 2^-24) and the low half, scaled by 2048, resolves v to 2^-24 / 2048 / 2 = 2^-36: the
    error of a product then has an ABSOLUTE floor of ~2^-36 |other operand| per term instead of a relative 2^-21.  The bound the kernels
    document is therefore  |err| <= 2^-20 sum|x w| + K 2^-35 max|w|  (x the small operand): asserted here for x of magnitude 1e-3 ... 3e-7
    against fp64, next to the fp32 MFMA form - which has no such floor.  (The model's operands - edge embeddings, relu activations, BatchNorm-ed
    node features - are O(1e-2 ... 10); the parity suites and the fp64 comparison of all size classes run on them.)"""
    import ctypes as C
    from diffdock_pocket_amd import _lib as L
    from diffdock_pocket_amd.packing import split_h2
    from diffdock_pocket_amd.score_model import _stream
    torch.manual_seed(3)
    dev = _dev()
    lib = L.load()
    k, ncols, nrows, ldx = 60, 2560, 300, 184
    x = (torch.randn(nrows, ldx) * mag).to(dev)
    w = (torch.randn(1, k, ncols) * 0.3).to(dev)
    offs_c = (C.c_int32 * 1)(0)
    wh = split_h2(w)
    exact = x[:, :k].double() @ w[0].double()
    scale = x[:, :k].double().abs() @ w[0].double().abs()
    outh = torch.empty(1, nrows, ncols, device=dev)
    L.check(lib.ddp_stage_a_h2(x.data_ptr(), ldx, nrows, None, None, nrows, offs_c, 1, w.data_ptr(), wh.data_ptr(), k, ncols, outh.data_ptr(),
                               ncols, None, _stream()), "ddp_stage_a_h2")
    err = (outh[0].double() - exact).abs()
    bound = 2.0 ** -20 * scale + k * 2.0 ** -35 * float(w.abs().max())
    assert bool((err <= bound).all()), (mag, float((err / bound).max()))
    if mag <= 1e-5:     # (the floor is real: the relative bound alone does not hold down there)
        assert float((err / scale).max()) > 2.0 ** -20


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("mag_w", [0.3, 0.02])
def test_stage_a_plane_forms_against_fp64(fmt, mag_w):
    """ddp_stage_a_gh / ddp_stage_a_gh3 by themselves: the G rows they write, decoded from the BYTES by the header's description of the two
    plane forms (tests/helpers.decode_gh_rows), against an fp64 product of the same right-hand sides.
      form 0 (hi + lo fp16 words): the unified planes of both operands carry 22 bits, the planes written carry 22: |err| <= 2^-20 sum|x w|
        + the absolute floor 2^-25 of a subnormal lo word;
      form 1 (hi fp16 truncated + a continuation byte: 19 significant bits, ABI 17): |err| <= 2^-19 |V| + 2^-24 (below the fp16 normal range
        the hi word alone) + the product's own 2^-20 sum|x w|.
    mag_w = 0.02: the fc.3 x block-scale magnitudes of ADVICE round 5 (the right-hand side's planes are those of 16 w since DDP_GH_SX = 2:
    lo words stay normal).  Values are V = DDP_ROWS_SG G; Gb columns are fp32 in both forms."""
    import ctypes as C
    from diffdock_pocket_amd import _lib as L
    from diffdock_pocket_amd import packing as P
    from diffdock_pocket_amd.score_model import _stream
    from helpers import decode_gh_rows
    torch.manual_seed(11 + fmt)
    dev = _dev()
    lib = L.load()
    hid, widths, k, nrows, ldx = 180, [32, 28, 12], 60, 333, 184
    n8, gcp = 23, sum(widths)
    ld = P.gh_ld(hid, gcp) if fmt != 1 else P.gh3_ld(hid, gcp)
    ncols = ld if fmt != 1 else ld // 6 * 8
    # right-hand side in the product's column order: plane groups [part][k8][c][8], then the Gb columns (form 1: six per 8-column group)
    Wv = torch.randn(k, n8, gcp, 8) * mag_w * 32.0                 # (the plane scale rides in the columns)
    Wb = torch.randn(k, gcp) * mag_w * 512.0
    W = torch.zeros(k, ncols)
    cum = 0
    for w_ in widths:
        W[:, 8 * n8 * cum:8 * n8 * (cum + w_)] = Wv[:, :, cum:cum + w_].reshape(k, -1)
        cum += w_
    c = torch.arange(gcp)
    W[:, (8 * n8 * gcp + c) if fmt != 1 else (8 * n8 * gcp + 8 * (c // 6) + torch.tensor(P.GH3_FP32_COLS)[c % 6])] = Wb
    x = torch.randn(nrows, ldx)
    xd, Wd = x.to(dev), W.unsqueeze(0).contiguous().to(dev)
    wh = P.split_h2(Wd, unified_scale=P.GH_SW)
    dest = P.gh_dest_table(widths, n8, ncols, fmt=fmt).unsqueeze(0).contiguous().to(dev)
    out = torch.full((1, nrows, ld), float("nan"), device=dev)
    offs_c = (C.c_int32 * 1)(60)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    fn = lib.ddp_stage_a_gh3 if fmt == 1 else lib.ddp_stage_a_gh
    L.check(fn(xd.data_ptr(), ldx, nrows, None, None, nrows, offs_c, 1, Wd.data_ptr(), wh.data_ptr(), k, ncols, out.data_ptr(), ld, flag.data_ptr(),
               dest.data_ptr(), _stream()), "ddp_stage_a_gh")
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    V, Gb = decode_gh_rows(out[0], widths, hid, fmt)
    xs = x[:, 60:60 + k].double()
    exactV = torch.einsum("nk,kgcs->ngcs", xs, Wv.double())
    scaleV = torch.einsum("nk,kgcs->ngcs", xs.abs(), Wv.double().abs())
    exactB, scaleB = xs @ Wb.double(), xs.abs() @ Wb.double().abs()
    assert bool(torch.isfinite(V).all())
    errV = (V - exactV).abs()
    bound = 2.0 ** -20 * scaleV + (2.0 ** -25 if fmt != 1 else 2.0 ** -19 * exactV.abs() + 2.0 ** -24)
    assert bool((errV <= bound).all()), (fmt, float((errV / bound).max()))
    assert bool(((Gb - exactB).abs() <= 2.0 ** -20 * scaleB).all())
    if fmt == 1:      # (the continuation byte is what it claims to be: clearly coarser than two fp16 words)
        big = exactV.abs() > 1.0
        assert float((errV[big] / exactV.abs()[big]).max()) > 2.0 ** -22



@pytest.mark.parametrize("fmt,limit", [(0, 65504.0), (1, 65504.0)])
def test_stage_a_plane_forms_report_values_outside_their_range(fmt, limit):
    """A plane value the forms cannot hold - |V| > 65504, the hi word's range in both - raises range_flag (pinned host memory in the product: the forward then reruns in the fp32 form); just inside
    the limit it stays 0."""
    import ctypes as C
    from diffdock_pocket_amd import _lib as L
    from diffdock_pocket_amd import packing as P
    from diffdock_pocket_amd.score_model import _stream
    dev = _dev()
    lib = L.load()
    hid, widths, k, nrows, ldx = 180, [32, 28, 12], 60, 64, 64
    n8, gcp = 23, sum(widths)
    ld = P.gh_ld(hid, gcp) if fmt != 1 else P.gh3_ld(hid, gcp)
    ncols = ld if fmt != 1 else ld // 6 * 8
    W = torch.zeros(1, k, ncols)
    W[0, 0, :8 * n8 * gcp] = 4.0                       # V = 4 x[:, 0] in every plane group (x itself stays inside ITS split's range)
    Wd = W.to(dev)
    wh = P.split_h2(Wd, unified_scale=P.GH_SW)
    dest = P.gh_dest_table(widths, n8, ncols, fmt=fmt).unsqueeze(0).contiguous().to(dev)
    fn = lib.ddp_stage_a_gh3 if fmt == 1 else lib.ddp_stage_a_gh
    for value, want in ((0.97 * limit, 0), (1.03 * limit, 1)):
        x = torch.zeros(nrows, ldx)
        x[5, 0] = value / 4.0
        xd = x.to(dev)
        out = torch.empty((1, nrows, ld), device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        L.check(fn(xd.data_ptr(), ldx, nrows, None, None, nrows, (C.c_int32 * 1)(0), 1, Wd.data_ptr(), wh.data_ptr(), k, ncols, out.data_ptr(), ld,
                   flag.data_ptr(), dest.data_ptr(), _stream()), "ddp_stage_a_gh")
        torch.cuda.synchronize()
        assert int(flag.item()) == want, (fmt, value, int(flag.item()))


def test_occupancy_shaping_changes_no_bit():
    """ddp_set_occupancy_shaping (ABI 16) only changes how many workgroups of ddp_conv_rows / stage A share a CU (launch-time LDS floors,
    profiles/r06_overlap_ab.txt): the same kernels with the same arguments - the forward's scores are bit for bit the unshaped ones, and
    (0, 0) restores the kernels' own occupancy."""
    from diffdock_pocket_amd import launch as K
    case, gold, batch, sd = case_inputs("cfg2_noflex")
    model = _model_for(case, sd)
    b = batch.to(_dev())
    base = [t.clone() for t in model(b)]
    K.occupancy_shaping(82 * 1024, 30 * 1024)
    try:
        shaped = [t.clone() for t in model(b)]
    finally:
        K.occupancy_shaping(0, 0)
    again = [t.clone() for t in model(b)]
    for a, s_, c in zip(base, shaped, again):
        assert torch.equal(a, s_) and torch.equal(a, c)
    with pytest.raises(Exception):
        K.occupancy_shaping(-1, 0)


def test_forward_with_bf16x3_stage_a_agrees_with_the_exact_form():
    """model.stage_a_bf16x3 (an option, off by default): the same forward with stage A as bf16x3 products - scores within 2e-5
    of the exact-fp32 form's (parity tolerance of the path: 1e-4)."""
    from diffdock_pocket_amd import launch as K
    case, gold, batch, sd = case_inputs("cfg2_noflex")
    model = _model_for(case, sd)
    b = batch.to(_dev())
    out = {}
    rows_was = K.CONV_ROWS
    K.CONV_ROWS = False      # (the option belongs to the 32-edge kernel's fp32 G layout; ddp_conv_rows reads G in plane form from ddp_stage_a_gh)
    try:
        for flag in (False, True):
            model.stage_a_bf16x3 = flag
            out[flag] = [t.clone() for t in model(b)]
    finally:
        K.CONV_ROWS = rows_was
    assert any(not torch.equal(a, c) for a, c in zip(out[False], out[True]))      # (the option did switch kernels)
    for a, c in zip(out[False], out[True]):
        assert elementwise_excess(c, a, 2e-5, 2e-5) <= 1


@pytest.mark.parametrize("n_list,cap", [(37, 300), (300, 300), (0, 64), (9000, 20000)])
def test_stage_a_row_list_with_a_device_side_length(n_list, cap):
    """ddp_stage_a on a ROW LIST whose length lives in device memory: exactly the listed rows of out are written (in place, at
    their own row index), bit for bit what the dense product gives for them; the rest of out is untouched."""
    import ctypes as C
    from diffdock_pocket_amd import _lib as L
    from diffdock_pocket_amd.score_model import _stream
    torch.manual_seed(n_list + cap)
    dev = _dev()
    lib = L.load()
    k, ncols, nrows, ldx = 60, 12600, max(cap, 1), 180
    nb, offs = 2, (120, 0)
    ldo = (ncols + 31) // 32 * 32
    x, w = torch.randn(nrows, ldx, device=dev), torch.randn(nb, k, ncols, device=dev)
    dense = torch.zeros((nb, nrows, ldo), device=dev)
    offs_c = (C.c_int32 * nb)(*offs)
    L.check(lib.ddp_stage_a(x.data_ptr(), ldx, nrows, None, None, nrows, offs_c, nb, w.data_ptr(), None, k, ncols, dense.data_ptr(), ldo,
                            _stream()), "ddp_stage_a")
    rows = torch.randperm(nrows, device=dev)[:cap].to(torch.int32).sort().values.contiguous()
    n_dev = torch.tensor([n_list], dtype=torch.int32, device=dev)
    out = torch.full((nb, nrows, ldo), -7.0, device=dev)
    L.check(lib.ddp_stage_a(x.data_ptr(), ldx, cap, rows.data_ptr(), n_dev.data_ptr(), nrows, offs_c, nb, w.data_ptr(), None, k, ncols,
                            out.data_ptr(), ldo, _stream()), "ddp_stage_a")
    torch.cuda.synchronize()
    listed = rows[:n_list].long()
    mask = torch.zeros(nrows, dtype=torch.bool, device=dev)
    mask[listed] = True
    assert torch.equal(out[:, mask, :ncols], dense[:, mask, :ncols])
    assert bool((out[:, ~mask] == -7.0).all()) and bool((out[:, :, ncols:] == -7.0).all())


def test_sde_update_and_replicated_reduce():
    """ddp_sde_update against the PyTorch expression `a * score + b * z` (separate roundings) and ddp_segment_reduce's
    replicated add (n_rep) against reduce-then-add."""
    import ctypes as C
    from diffdock_pocket_amd import _lib as L
    from diffdock_pocket_amd import graph as G
    from diffdock_pocket_amd import launch as K
    from diffdock_pocket_amd.score_model import _stream
    torch.manual_seed(0)
    dev = _dev()
    lib = L.load()
    coef = torch.randn(8, device=dev)
    shapes = [(40, 3), (40, 3), (40, 7), (40, 18)]
    sc = [torch.randn(sh, device=dev) for sh in shapes]
    zz = [torch.randn(sh, device=dev) for sh in shapes]
    out = [torch.empty(sh, device=dev) for sh in shapes]
    a = L.SdeArgs()
    for k in range(4):
        a.score[k], a.z[k], a.out[k], a.n[k] = sc[k].data_ptr(), zz[k].data_ptr(), out[k].data_ptr(), sc[k].numel()
    L.check(lib.ddp_sde_update(coef.data_ptr(), C.byref(a), _stream()), "ddp_sde_update")
    for k in range(4):
        assert torch.equal(out[k], float(coef[2 * k]) * sc[k] + float(coef[2 * k + 1]) * zz[k])
    # replicated reduce: the update of n0 nodes (computed from 0) added to the rows of B graphs
    B, n0, d, E = 5, 30, 36, 400
    recv = torch.sort(torch.randint(0, n0, (E,))).values
    csr = G.build_csr(recv.to(dev), torch.zeros(E, dtype=torch.long, device=dev), n0, presorted=True)
    msg = torch.randn(E, d, device=dev)
    pk = type("PK", (), {"bn_scale": torch.rand(d, device=dev) + 0.5, "bn_shift": torch.randn(d, device=dev)})()
    x = torch.randn(B * n0, 40, device=dev)
    want = x.clone()
    u0 = torch.zeros(n0, d, device=dev)
    K.launch_reduce(u0, d, n0, d, [(msg, csr, pk)], accumulate=False)
    want.view(B, n0, 40)[:, :, :d].add_(u0)
    K.launch_reduce(x, 40, n0, d, [(msg, csr, pk)], accumulate=True, n_rep=B, rep_stride=n0)
    assert torch.equal(x, want)


def test_static_graph_cache_is_invalidated_by_in_place_updates():
    """The receptor-side graph (atom kNN, CSR views) is kept across forward calls while its input tensors are unchanged
    (score_model._cached).  Moving only the ligand must reuse it, moving atoms in place (side-chain updates) or handing
    over a new batch must rebuild it: every call has to equal a fresh model's output bitwise."""
    case, gold, batch, sd = case_inputs("cfg2_small")
    dev = _dev()
    cached, b = _model_for(case, sd), case.make_batch().to(dev)

    def fresh(bb):
        return [t.clone() for t in _model_for(case, sd)(bb)]

    def same(x, y):
        return all(torch.equal(p, q) for p, q in zip(x, y))

    out0 = [t.clone() for t in cached(b)]
    assert same(out0, fresh(b))
    n_hit = len(cached._static_cache)
    assert n_hit > 0
    b["ligand"].pos = b["ligand"].pos + 0.3                       # ligand moves: the static entries are reused
    keys_before = {k: id(v[3]) for k, v in cached._static_cache.items() if k in ("aa", "c_aa", "c_rr")}
    out1 = [t.clone() for t in cached(b)]
    assert {k: id(v[3]) for k, v in cached._static_cache.items() if k in keys_before} == keys_before
    assert same(out1, fresh(b)) and not same(out1, out0)
    b["atom"].pos.add_(torch.randn_like(b["atom"].pos) * 0.4)     # atoms move IN PLACE: version counter bumps -> rebuild
    out2 = [t.clone() for t in cached(b)]
    assert id(cached._static_cache["aa"][3]) != keys_before["aa"]
    assert same(out2, fresh(b)) and not same(out2, out1)
    b2 = case.make_batch().to(dev)                                # a different batch object
    assert same([t.clone() for t in cached(b2)], out0)


def test_pose_update_kernel_matches_modify_conformer():
    """ddp_pose_update (one launch: rigid move, torsions, Kabsch re-alignment via Horn's quaternion) against the batched
    PyTorch modify_conformer, which tests/test_sampler_cpu.py pins to the reference's own functions."""
    import numpy as np
    from diffdock_pocket_amd import sampler as S
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    g = make_3dpf_complex(seed=0, flexible_sidechains=False)
    em = g["ligand"].edge_mask.bool()
    bonds = g["ligand", "ligand"].edge_index.t()[em].clone()
    mr = g["ligand"].mask_rotate
    mask = torch.as_tensor(np.asarray(mr if isinstance(mr, np.ndarray) else mr[0])).bool()
    T, n = bonds.shape[0], g["ligand"].pos.shape[0]
    torch.manual_seed(3)
    N = 17
    pos = g["ligand"].pos.unsqueeze(0).repeat(N, 1, 1) + torch.randn(N, 1, 3) * 3
    tr, rot, tor = torch.randn(N, 3) * 0.5, torch.randn(N, 3) * 0.4, torch.randn(N, T) * 0.6
    rot[0] = 0.0                                                   # zero rotation vector: small-angle branch
    tor[1] = 0.0
    want = S.modify_conformer(pos.double(), tr.double(), rot.double(), tor.double(), bonds, mask)
    got = S.modify_conformer_hip(pos.to(dev), tr.to(dev), rot.to(dev), tor.to(dev), bonds.to(torch.int32).to(dev),
                                 mask.to(torch.uint8).to(dev)).cpu().double()
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())
    want_r = S.modify_conformer(pos.double(), tr.double(), rot.double(), None, bonds, mask)
    got_r = S.modify_conformer_hip(pos.to(dev), tr.to(dev), rot.to(dev), None, None, None).cpu().double()
    assert float((got_r - want_r).abs().max()) < 2e-5 * float(want_r.abs().max())


def _ragged_points(sizes, scale, seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.cat([torch.randn(n, 3, generator=g) * scale + 3.0 * i for i, n in enumerate(sizes)])
    batch = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(sizes)])
    return pos, batch


@pytest.mark.parametrize("rule", ["first_index", "nearest"])
@pytest.mark.parametrize("cap", [10000, 32, 5])
def test_hip_radius_matches_dense_formulation(cap, rule):
    """ddp_radius_count / ddp_radius_fill (csrc/ddp_graph.hip) against the dense PyTorch formulation of graph.py, which
    tests/test_host_logic.py pins to the oracle's torch_cluster restatement: identical pairs in identical order, ragged
    graphs (one of them empty on the x side), cap binding and not binding, self-loop removal after the cap."""
    from diffdock_pocket_amd import graph as G
    dev = _dev()
    x, bx = _ragged_points([40, 7, 0, 120, 33], 2.0, 1)
    y, by = _ragged_points([9, 3, 5, 30, 1], 2.0, 2)
    B = 5
    lx_c, ly_c = G.DenseLayout.build(bx, B), G.DenseLayout.build(by, B)
    lx_d, ly_d = G.DenseLayout.build(bx.to(dev), B), G.DenseLayout.build(by.to(dev), B)
    for r in (1.5, 3.0):
        want = G.radius(x, y, r, lx_c, ly_c, max_num_neighbors=cap, truncation=rule)
        got = G.radius(x.to(dev), y.to(dev), r, lx_d, ly_d, max_num_neighbors=cap, truncation=rule).cpu()
        assert want.shape == got.shape and torch.equal(want, got), (r, cap, want.shape, got.shape)
        want_g = G.radius_graph(x, r, lx_c, max_num_neighbors=cap, truncation=rule)
        got_g = G.radius_graph(x.to(dev), r, lx_d, max_num_neighbors=cap, truncation=rule).cpu()
        assert torch.equal(want_g, got_g), (r, cap)


@pytest.mark.parametrize("k", [1, 8, 32])
def test_hip_knn_matches_dense_formulation(k):
    from diffdock_pocket_amd import graph as G
    dev = _dev()
    for sizes in ([50, 50, 50], [40, 3, 1, 120, 9]):      # uniform and ragged (graphs smaller than k + 1)
        x, bx = _ragged_points(sizes, 2.0, 3)
        lc, ld = G.DenseLayout.build(bx, len(sizes)), G.DenseLayout.build(bx.to(dev), len(sizes))
        want = G.knn_graph(x, k, lc)
        got = G.knn_graph(x.to(dev), k, ld).cpu()
        assert torch.equal(want, got), (k, sizes, want.shape, got.shape)


def test_sidechain_update_kernel_matches_pytorch_form():
    from diffdock_pocket_amd import sampler as S
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    g = make_3dpf_complex(seed=0, flexible_sidechains=True)
    fr = g["flexResidues"]
    torch.manual_seed(5)
    N = 6
    pos = g["atom"].pos.unsqueeze(0).repeat(N, 1, 1) + torch.randn(N, 1, 3)
    ang = torch.randn(N, fr.edge_idx.shape[0]) * 0.8
    want = S.apply_sidechain_torsions(pos.double(), fr.edge_idx, fr.subcomponents, fr.subcomponentsMapping, ang.double())
    got = S.apply_sidechain_torsions_hip(pos.to(dev), fr.edge_idx.to(torch.int32).to(dev), fr.subcomponents.to(torch.int32).to(dev),
                                         fr.subcomponentsMapping.to(torch.int32).to(dev), ang.to(dev)).cpu().double()
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())
    assert float((got - pos.double()).abs().max()) > 0.1          # something moved


@pytest.mark.parametrize("name", ["cfg2_noflex", "cfg2_small"])
def test_layer0_sharing_across_identical_receptors(name):
    """A sampling batch = N poses of one complex at one diffusion time: the layer-0 receptor-side convs are computed for
    graph 0 only (score_model.share_layer0): all four of them for a rigid receptor, receptor<-receptor alone when side chains
    are flexible (they move per sample).  Same result as the general path (up to the changed summation order of the residual
    update) and as the CPU oracle; a batch with different times must not take the shortcut."""
    from diffdock_pocket_amd.batch import collate, set_time
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    case, gold, batch, sd = case_inputs(name)
    dev = _dev()
    model = _model_for(case, sd)
    gs = []
    g = torch.Generator().manual_seed(11)
    for i in range(3):
        c = make_3dpf_complex(seed=case.data_seed, flexible_sidechains=case.flexible_sidechains, n_rec=case.n_rec)
        c["ligand"].pos = c["ligand"].pos + torch.randn(1, 3, generator=g) * 1.5
        gs.append(c)
    b = collate(gs)
    set_time(b, 0.4, 0.4, 0.4, 0.4)
    bd = b.to(dev)
    model.share_layer0 = True
    model.flex_share_min_work = 0
    fast = [t.clone() for t in model(bd)]
    if case.flexible_sidechains:
        # (+ "flex": features and atom-receptor edges repeat, so layer 0's atom side is shared wherever no atom moved - here
        # every copy has the same side-chain pose: samples 1, 2 read everything from sample 0, test_flexible_layer0_sharing_is_exact)
        assert {6, "flex"} == {k for k, v in model._static_cache["shared0_rec"][3].items() if v is not None}
        assert model.last_stats["flex0_kept_aa_edges"] == model.last_stats["E_aa"] // 3
    else:
        assert {3, 5, 6, 8} <= {k for k, v in model._static_cache["shared0"][3].items() if v is not None}
    model.share_layer0 = False
    slow = [t.clone() for t in model(bd)]
    want = OracleScoreModel(case.oracle_config(), sd)(b)
    for f, s_, w in zip(fast, slow, want):
        if w.numel():
            assert rel_err(f.float().cpu(), s_.float().cpu()) < 2e-6
            assert rel_err(f.float().cpu(), w) < TOL
    # different times per graph: the general path is taken and gives the general result
    model.share_layer0 = True
    set_time(bd, 0.4, 0.4, 0.4, 0.4)
    bd["receptor"].node_t["tr"][0] += 0.1
    bd["atom"].node_t["tr"][-1] += 0.1
    mixed = [t.clone() for t in model(bd)]
    model.share_layer0 = False
    mixed_ref = [t.clone() for t in model(bd)]
    for x, y in zip(mixed, mixed_ref):
        assert torch.equal(x, y)


def test_layer1_clean_pair_sharing_is_exact():
    """score_model.share_clean_layer1: in a sampling batch of one rigid complex the layer-1 atom<-atom messages between atoms
    that no ligand message has reached are computed once and read through ddp_segment_reduce's row map.  Bitwise the general
    path (same per-edge arithmetic, same summation order), and actually taken: most of the edges are shared."""
    from diffdock_pocket_amd.batch import collate, set_time
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    case, gold, batch, sd = case_inputs("cfg2_noflex")          # ns=60 nv=10 L=4, rigid receptor
    dev = _dev()
    model = _model_for(case, sd)
    gs = []
    g = torch.Generator().manual_seed(4)
    for i in range(5):
        c = make_3dpf_complex(seed=case.data_seed, flexible_sidechains=False, n_rec=40)
        c["ligand"].pos = c["ligand"].pos + torch.randn(1, 3, generator=g) * 2.0
        gs.append(c)
    b = collate(gs)
    set_time(b, 0.6, 0.6, 0.6, 0.6)
    bd = b.to(dev)
    model.share_clean_layer1 = True
    fast = [t.clone() for t in model(bd)]
    st = dict(model.last_stats)
    assert "clean1_dirty_edges" in st and 0 < st["clean1_dirty_edges"] < 0.6 * st["E_aa"], st
    model.share_clean_layer1 = False
    slow = [t.clone() for t in model(bd)]
    assert "clean1_dirty_edges" not in model.last_stats
    for f, s_ in zip(fast, slow):
        assert torch.equal(f, s_)
    want = OracleScoreModel(case.oracle_config(), sd)(b)
    for f, w in zip(fast, want):
        if w.numel():
            assert rel_err(f.float().cpu(), w) < TOL
    # ligands far away: every atom is clean, nothing is computed per sample
    b2 = collate(gs)
    b2["ligand"].pos = b2["ligand"].pos + 30.0
    set_time(b2, 0.6, 0.6, 0.6, 0.6)
    model.share_clean_layer1 = True
    far = [t.clone() for t in model(b2.to(dev))]
    assert model.last_stats["clean1_dirty_edges"] == 0
    model.share_clean_layer1 = False
    for f, s_ in zip(far, model(b2.to(dev))):
        assert torch.equal(f, s_)


@pytest.mark.parametrize("n", [6, 40])
def test_flexible_layer0_sharing_is_exact(n):
    """score_model.share_flex_layer0 (flexible side chains: N poses of one complex whose side chains differ): layer 0's
    atom<-atom / atom<-receptor / receptor<-atom messages are computed for sample 0 and, in the other samples, only for the
    receivers with a moved atom on an incoming edge or in reach of their kNN list; everyone else reads sample 0's messages through
    the segmented mean's row map, and layer 1's clean-pair sharing works on top with those receivers counted as touched.
    Bitwise the general path - at the first step (every side chain freshly randomised) and after two denoising steps."""
    import bench
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    sched = get_t_schedule(20)
    g = make_3dpf_complex(seed=0, flexible_sidechains=True)
    model, kw = bench.build_model("cfg2", True, dev)
    smp = Sampler(model, g, n, dev, SamplerConfig(inference_steps=20, flexible_sidechains=True, hip_graph=False), seed=0)
    smp.randomize()
    for rnd in range(2):
        model.share_flex_layer0 = True
        fast = [t.clone() for t in smp.scores(float(sched[2 * rnd]))]
        st = dict(model.last_stats)
        assert 0 < st["flex0_kept_aa_edges"] < 0.7 * st["E_aa"], st      # sample 0 + the dirty receivers of the others
        assert "clean1_dirty_edges" in st and st["clean1_dirty_edges"] < st["E_aa"], st
        model.share_flex_layer0 = False
        slow = [t.clone() for t in smp.scores(float(sched[2 * rnd]))]
        assert "flex0_kept_aa_edges" not in model.last_stats and "clean1_dirty_edges" not in model.last_stats
        for f, s_ in zip(fast, slow):
            assert torch.equal(f, s_), rel_err(f, s_)
        model.share_flex_layer0 = True
        smp.step(2 * rnd, sched)
        smp.step(2 * rnd + 1, sched)


@pytest.mark.parametrize("name", ["cfg2_noflex", "cfg2_small", "cfg1_full", "ns24_l3"])
def test_last_receptor_layer_pruning_is_exact(name):
    """Dead-output elimination over the last layers (score_model.prune_last_receptor_layer): the receptor-side convs of
    the last layers keep only the edges that end in a node whose features are still read (without flexible side chains:
    layer L-2, read by the final ligand<-atom / ligand<-receptor convs; with them: layers L-1 and L-2, read by the
    side-chain torsion head).  Must not change a bit of the result."""
    case, gold, batch, sd = case_inputs(name)
    dev = _dev()
    model = _model_for(case, sd)
    b = case.make_batch().to(dev)
    model.prune_last_receptor_layer = True
    a = [t.clone() for t in model(b)]
    model.prune_last_receptor_layer = False
    c = [t.clone() for t in model(b)]
    for x, y in zip(a, c):
        assert torch.equal(x, y)


@pytest.mark.parametrize("flex", [False, True])
def test_sampler_end_to_end_on_device(flex):
    """The reverse-diffusion loop on the device: a few denoising steps of a small batch through the HIP score model and
    the HIP pose / side-chain updates, then the confidence pass; the same seeds on a second sampler reproduce the poses
    bitwise (deterministic kernels), and a sampler holding only the second half of the samples (a rank's shard) gets
    the poses of that half after one step (to fp32 rounding: a few batch-level PyTorch reductions depend on the batch
    composition; with random-init weights the score is not smooth, so later steps amplify that rounding and are not
    compared - the exact shard-invariance check is the 2-rank CPU test in tests/test_sampler_cpu.py)."""
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    case = CASES["cfg1_full"] if flex else CASES["cfg2_noflex"]
    _, _, _, sd = case_inputs(case.name)
    model = _model_for(case, sd)
    conf_case = CASES["conf_ns24_l5"] if flex else CASES["conf_noflex"]
    conf_model = _model_for(conf_case, case_inputs(conf_case.name)[3])
    g = make_3dpf_complex(seed=0, flexible_sidechains=flex, n_rec=16)
    sched = get_t_schedule(4)

    def run(n_total, sl=None, steps=4):
        s = Sampler(model, g, n_total, dev, SamplerConfig(inference_steps=4, flexible_sidechains=flex), seed=3, sample_slice=sl)
        s.randomize()
        for i in range(steps):
            s.step(i, sched)
        return s, s.lig_pos.clone(), s.atom_pos.clone()

    s_all, lig_all, atoms_all = run(4)
    assert torch.isfinite(lig_all).all() and torch.isfinite(atoms_all).all()
    _, lig_again, atoms_again = run(4)
    assert torch.equal(lig_all, lig_again) and torch.equal(atoms_all, atoms_again)
    _, lig_one, _ = run(4, steps=1)
    _, lig_half, _ = run(4, slice(2, 4), steps=1)
    assert torch.allclose(lig_half, lig_one[2:4], atol=5e-4, rtol=0)    # same noise stream; batch composition differs
    conf, order = s_all.confidence(conf_model)
    assert conf.shape[0] == 4 and sorted(order.tolist()) == [0, 1, 2, 3] and torch.isfinite(conf).all()
    # two resident groups stepped alternately on their own streams = the two shards run on their own, bit for bit
    from diffdock_pocket_amd.sampler import PipelinedSampler
    pipe = PipelinedSampler(model, g, 4, dev, SamplerConfig(inference_steps=4, flexible_sidechains=flex), seed=3, ways=2)
    pipe.randomize()
    for i in range(4):
        pipe.step(i, sched)
    lig_pipe, atoms_pipe = pipe.lig_pos.clone(), pipe.atom_pos.clone()
    _, lig_a, atoms_a = run(4, slice(0, 2))
    _, lig_b, atoms_b = run(4, slice(2, 4))
    assert torch.equal(lig_pipe, torch.cat([lig_a, lig_b])) and torch.equal(atoms_pipe, torch.cat([atoms_a, atoms_b]))
    assert model.cache_slot == 0
    # ... and through run(): snapshot, all steps, one range check, the overflow check - the same poses; with the range flag forced once the
    # job is put back to its snapshot and run again in the fp32 form (one recovery counted, every group restored: finite poses close to the
    # split form's - one step's fp32-vs-split difference on a random-init network, not a restart from somewhere else)
    pipe2 = PipelinedSampler(model, g, 4, dev, SamplerConfig(inference_steps=4, flexible_sidechains=flex), seed=3, ways=2)
    pipe2.randomize()
    lig_run, atoms_run = pipe2.run(sched)
    assert torch.equal(lig_run, lig_pipe) and torch.equal(atoms_run, atoms_pipe)
    if model.conv_h2:
        pipe3 = PipelinedSampler(model, g, 4, dev, SamplerConfig(inference_steps=4, flexible_sidechains=flex), seed=3, ways=2)
        pipe3.randomize()
        start = pipe3.snapshot()
        real = model.range_flag_raised
        fired = []

        def once(clear=True):
            if not fired:
                fired.append(1)
                return True
            return real(clear)
        model.range_flag_raised = once
        before = model.__dict__.get("h2_recoveries", 0)
        try:
            lig_rec, atoms_rec = pipe3.run(sched)
        finally:
            del model.__dict__["range_flag_raised"]
        assert model.__dict__.get("h2_recoveries", 0) == before + 1 and model.conv_h2
        assert torch.isfinite(lig_rec).all() and torch.isfinite(atoms_rec).all()
        # the rerun started from the snapshot: its first step from there reproduces, in the fp32 form, what a fresh sampler's first step gives
        pipe3.restore(start)
        pipe3.step(0, sched)
        _, lig_one_a, _ = run(4, slice(0, 2), steps=1)
        _, lig_one_b, _ = run(4, slice(2, 4), steps=1)
        assert torch.equal(pipe3.lig_pos, torch.cat([lig_one_a, lig_one_b]))
    if flex:
        assert float((atoms_all - g["atom"].pos.to(dev)).abs().max()) > 1e-3      # side chains moved


def test_three_step_trajectory_matches_the_cpu_oracle_trajectory():
    """A multi-step trajectory on the device (HIP score model + HIP pose / side-chain update kernels) against the same
    trajectory on the CPU with the ORACLE as score function and the PyTorch pose update (which tests/test_sampler_cpu.py pins
    to the reference's own loop): same seeded start, same noise stream.  cfg1 on the full complex, 3 of a 20-step schedule's
    first steps.  The synthetic weights are scaled down (x 0.5 on every conv fc layer) so that the random-init network is
    smooth enough for per-step fp32 rounding (~1e-6 relative per forward) not to be amplified beyond the tolerance."""
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    case = CASES["cfg1_full"]
    _, _, _, sd = case_inputs(case.name)
    sd = {k: (v * 0.5 if (".fc." in k and k.endswith("weight")) else v) for k, v in sd.items()}
    model = _model_for(case, sd)
    oracle = OracleScoreModel(case.oracle_config(), sd)
    g = make_3dpf_complex(seed=0, flexible_sidechains=True)
    sched = get_t_schedule(20)
    cfg = SamplerConfig(inference_steps=20, flexible_sidechains=True)
    s_gpu = Sampler(model, g, 4, dev, cfg, seed=9)
    s_cpu = Sampler(lambda b: oracle(b), g, 4, torch.device("cpu"), cfg, seed=9)
    s_gpu.randomize()
    s_cpu.randomize()
    assert float((s_gpu.lig_pos.cpu() - s_cpu.lig_pos).abs().max()) < 1e-4
    for i in range(3):
        s_gpu.step(i, sched)
        s_cpu.step(i, sched)
        dl = float((s_gpu.lig_pos.cpu() - s_cpu.lig_pos).abs().max())
        da = float((s_gpu.atom_pos.cpu() - s_cpu.atom_pos).abs().max())
        assert dl < 2e-3 and da < 2e-3, (i, dl, da)       # angstrom; the ligand has moved by several angstrom by then
    assert float((s_cpu.lig_pos - g["ligand"].pos).abs().max()) > 1.0


@pytest.mark.parametrize("seed", [9, 3])
def test_cfg1_job_end_to_end_against_the_cpu_sampler(seed):
    """BASELINE configs[0] END TO END: 3dpf, 4 samples x ALL 20 denoising steps, the cfg1 score model with flexible side chains and
    the UNSCALED synthetic weights.  HIP sampler (device-resident step, captured hipGraph from the third step on) against the CPU
    sampler driven by the oracle (whose loop tests/test_sampler_cpu.py pins to the reference's own utils/sampling.py), same seeded
    start and noise stream.  The ligands travel 23 - 25 A; measured divergence of the final poses (tools/traj_divergence.py, one
    MI355X): <= 1.8e-4 A on ligand atoms, <= 1.7e-5 A on pocket atoms.  Bounds: 2e-3 A / 2e-4 A at every step."""
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    case = CASES["cfg1_full"]
    _, _, _, sd = case_inputs(case.name)
    model = _model_for(case, sd)
    oracle = OracleScoreModel(case.oracle_config(), sd)
    g = make_3dpf_complex(seed=0, flexible_sidechains=True)
    sched = get_t_schedule(20)
    cfg = SamplerConfig(inference_steps=20, flexible_sidechains=True)
    s_gpu = Sampler(model, g, 4, dev, cfg, seed=seed)
    s_cpu = Sampler(lambda b: oracle(b), g, 4, torch.device("cpu"), cfg, seed=seed)
    s_gpu.randomize()
    s_cpu.randomize()
    start = s_cpu.lig_pos.clone()
    for i in range(20):
        s_gpu.step(i, sched)
        s_cpu.step(i, sched)
        dl = float((s_gpu.lig_pos.cpu() - s_cpu.lig_pos).abs().max())
        da = float((s_gpu.atom_pos.cpu() - s_cpu.atom_pos).abs().max())
        assert dl < 2e-3 and da < 2e-4, (seed, i, dl, da)
    assert bool(s_gpu._graph)                                               # steps 3.. were graph replays
    assert float((s_cpu.lig_pos - start).abs().max()) > 10.0               # the poses did travel
    s_gpu.check_overflow()


@pytest.mark.parametrize("flex", [False, True])
def test_cfg2_job_end_to_end_against_the_cpu_sampler(flex):
    """The headline model END TO END: 3dpf, 2 samples x ALL 20 denoising steps of the cfg2 score model (ns=60 nv=10 L=6: BASELINE
    configs[1] / [2]'s network, bench.py's seeded random-init weights, unscaled), rigid receptor and flexible side chains.  HIP sampler
    (ddp_conv_rows / ddp_stage_a_gh / the 64-edge kernel in their fp16 hi/lo form, captured hipGraph from the third step on; with flexible
    side chains the first 12 steps) against the
    CPU sampler driven by the oracle (reference utils/sampling.py:93-251 restated, pinned by tests/test_sampler_cpu.py), same seeded start
    and noise stream, pose for pose at every step.  Measured (profiles/r05_cfg2_traj_divergence.txt): the per-step bounds below hold with
    a factor > 5 to spare; DDP_TRAJ_LOG=<file> appends the per-step divergences."""
    import bench
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    dev = _dev()
    model, kw = bench.build_model("cfg2", flex, dev)
    ocfg = OracleConfig(ns=kw["ns"], nv=kw["nv"], num_conv_layers=kw["num_conv_layers"], sigma_embed_dim=kw["sigma_embed_dim"],
                        distance_embed_dim=kw["distance_embed_dim"], cross_distance_embed_dim=kw["cross_distance_embed_dim"],
                        flexible_sidechains=kw["flexible_sidechains"], embedding_scale=1000.0)
    oracle = OracleScoreModel(ocfg, {k: v.detach().cpu() for k, v in model.state_dict().items()})
    g = make_3dpf_complex(seed=0, flexible_sidechains=flex)
    sched = get_t_schedule(20)
    cfg = SamplerConfig(inference_steps=20, flexible_sidechains=flex)
    s_gpu = Sampler(model, g, 2, dev, cfg, seed=5)
    s_cpu = Sampler(lambda b: oracle(b), g, 2, torch.device("cpu"), cfg, seed=5)
    s_gpu.randomize()
    s_cpu.randomize()
    start = s_cpu.lig_pos.clone()
    log = []
    # (flexible side chains: the first 12 of the 20 steps - the ligand has travelled > 5 A by then, the CPU oracle takes ~5 s per step and
    # the rigid job runs all 20; the suite's time, profiles/r06_pytest_gpu.txt)
    n_steps = 12 if flex else 20
    with torch.no_grad():
        for i in range(n_steps):
            s_gpu.step(i, sched)
            s_cpu.step(i, sched)
            dl = float((s_gpu.lig_pos.cpu() - s_cpu.lig_pos).abs().max())
            da = float((s_gpu.atom_pos.cpu() - s_cpu.atom_pos).abs().max())
            log.append((i, dl, da, float((s_cpu.lig_pos - start).abs().max())))
            assert dl < 2e-3 and da < 5e-4, (flex, i, dl, da)
    if os.environ.get("DDP_TRAJ_LOG"):
        with open(os.environ["DDP_TRAJ_LOG"], "a") as f:
            f.write(f"cfg2 2 samples x 20 steps, flexible_sidechains={flex}: step, max |ligand pose diff| (A), max |pocket atom diff| (A), ligand travel so far (A)\n")
            for row in log:
                f.write("  %2d  %.3e  %.3e  %.2f\n" % row)
    assert bool(s_gpu._graph)                                               # steps 3.. were graph replays
    assert float((s_cpu.lig_pos - start).abs().max()) > 5.0                # the poses did travel
    assert model.__dict__.get("h2_recoveries", 0) == 0
    s_gpu.check_overflow()
    s_gpu.close()


def test_csv_driver_on_device(tmp_path):
    """BASELINE configs[3] in miniature: inference.run_csv over a two-row csv (the reference's example complex, with and
    without flexible side chains, ESM rows from a dict) with the HIP score model and the HIP confidence model: poses are ranked
    by confidence, and the run is reproducible bit for bit."""
    import os
    from diffdock_pocket_amd import inference as INF
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    csv_path = tmp_path / "c.csv"
    csv_path.write_text("complex_name,experimental_protein,ligand,pocket_center_x,pocket_center_y,pocket_center_z,flexible_sidechains\n"
                        "a,3dpf_protein.pdb,3dpf_ligand.sdf,,,,A:160-A:193-A:197-A:198\n"
                        "b,3dpf_protein.pdb,3dpf_ligand.sdf\n")
    dev = _dev()
    case = CASES["cfg1_full"]
    model = _model_for(case, case_inputs(case.name)[3])
    conf_case = CASES["conf_ns24_l5"]
    conf_model = _model_for(conf_case, case_inputs(conf_case.name)[3])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        n_res = INF.build_row_graph(INF.load_protein_ligand_csv(str(csv_path))[0], root=gdir, allow_zero_esm=True)["receptor"].x.shape[0]
    esm = {k: torch.randn(n_res, 1280, generator=torch.Generator().manual_seed(i)) for i, k in enumerate("ab")}

    def run():
        return INF.run_csv(str(csv_path), model, dev, confidence_model=conf_model, samples_per_complex=6, inference_steps=4,
                           esm_embeddings=esm, root=gdir, seed=1)

    r1, r2 = run(), run()
    for a, b in zip(r1, r2):
        assert a.skipped is None and a.ligand_pos.shape[0] == 6 and torch.isfinite(a.ligand_pos).all()
        assert torch.equal(a.ligand_pos, b.ligand_pos) and torch.equal(a.order, b.order)
        key = a.confidence[:, 0] if a.confidence.dim() == 2 else a.confidence
        assert bool((key[:-1] >= key[1:]).all())


def test_forward_on_graph_from_the_input_pipeline():
    """a complex graph built from PDB / SDF text (inputs.build_complex_graph, the reference's example
    complex with the README's flexible residues) goes through the HIP forward and matches the oracle on the same batch."""
    import os
    from diffdock_pocket_amd import inputs as I
    from diffdock_pocket_amd.batch import collate, set_time
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = I.build_complex_graph(open(os.path.join(gdir, "3dpf_protein.pdb")).read(), open(os.path.join(gdir, "3dpf_ligand.sdf")).read(),
                              flexible_sidechains="A:160-A:193-A:197-A:198-A:222-A:224-A:227")
    gen = torch.Generator().manual_seed(11)
    g["receptor"].x = torch.cat([g["receptor"].x, torch.randn(g["receptor"].x.shape[0], 1280, generator=gen)], 1)
    graphs = []
    for shift in (0.0, 1.5):
        c = g.clone()
        c["ligand"].pos = c["ligand"].pos + shift * torch.randn(1, 3, generator=gen)
        graphs.append(c)

    def batch():
        b = collate(graphs)
        set_time(b, 0.7, 0.7, 0.7, 0.7)
        return b

    case, _, _, sd = case_inputs("cfg1_full")
    want = OracleScoreModel(case.oracle_config(), sd)(batch())
    got = _model_for(case, sd)(batch().to(_dev()))
    for a, w, k in zip(got, want, ("tr", "rot", "tor", "sc_tor")):
        assert a.shape == w.shape and w.numel() > 0, k
        assert rel_err(a.float().cpu(), w) < TOL, (k, rel_err(a.float().cpu(), w))


@pytest.mark.parametrize("E,n_recv,n_src", [(0, 7, 5), (1, 1, 1), (5000, 300, 40), (40000, 45000, 1500), (93000, 5600, 1480)])
def test_group_by_key_views_match_the_pytorch_definition(E, n_recv, n_src):
    """ddp_group_by_key (CSR and source-order views on the device) against graph.py's PyTorch stable-sort definition, bit
    for bit: ragged rows, empty rows, long rows (every item on one key), presorted input, empty edge set."""
    from diffdock_pocket_amd import graph as G
    dev = _dev()
    gen = torch.Generator().manual_seed(E + n_recv)
    recv = torch.randint(0, n_recv, (E,), generator=gen)
    src = torch.randint(0, n_src, (E,), generator=gen)
    if E >= 5000:
        recv[: E // 50] = n_recv // 2          # one long row (its items must come out in edge order)
    for presorted in (False, True):
        r = torch.sort(recv).values if presorted else recv
        want = G.build_csr(r, src, n_recv, presorted=presorted)
        got = G.build_csr(r.to(dev), src.to(dev), n_recv, presorted=presorted)
        got32 = G.build_csr(r.to(dev).to(torch.int32), src.to(dev).to(torch.int32), n_recv, presorted=presorted)
        for g in (got, got32):
            assert g.n_edges == want.n_edges
            for f in ("recv", "src", "eid", "rowptr"):
                a, b = getattr(g, f), getattr(want, f)
                assert a.dtype == torch.int32 and torch.equal(a.cpu(), b), (f, presorted)
        so_want = G.source_order(want)
        so_got = G.source_order(got, n_src)
        for f in ("recv", "src", "eid", "pos"):
            assert torch.equal(getattr(so_got, f).cpu(), getattr(so_want, f)), (f, presorted)
    again = G.source_order(G.build_csr(recv.to(dev), src.to(dev), n_recv), n_src)
    assert torch.equal(again.pos, G.source_order(G.build_csr(recv.to(dev), src.to(dev), n_recv), n_src).pos)
