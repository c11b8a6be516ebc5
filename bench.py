#!/usr/bin/env python3
"""bench.py - ligand poses/s of the reverse-diffusion hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--samples 40] [--flex] [--cfg cfg2|cfg1]

One "step" = one denoising step over the batch of sample graphs of the 3dpf complex: graph construction + score-model
forward (HIP) + SDE pose update.  Default workload = BASELINE.json configs[1]: 3dpf, 40 samples, full score model
(ns=60 nv=10, 6 conv layers), 20-step schedule; K steps walk the 20-step schedule cyclically, so the default K=20 is exactly
one 40-sample x 20-step job.  poses/s = samples * K / 20 / seconds (one pose = one sample carried through 20 steps),
aggregated over ranks.

Ranks.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N children (one per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set, a free port on 127.0.0.1) BEFORE anything touches the GPU, waits for them and
exits non-zero if any fails - the role of the reference's per-device process pool, inference.py:466-488.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the children exist already (WORLD_SIZE set) and
this process is one of them; a WORLD_SIZE that differs from --gpus is an error.  Backend "nccl" (= RCCL over xGMI); the only
collective on the data path is the all_gather of final ligand poses at the end of the timed region.
  --scaling weak   (default) every rank owns `--samples` samples of a job of N * samples samples (fixed work per GPU)
  --scaling strong the job has `--samples` samples in total, rank r owns [r*S/N, (r+1)*S/N) - BASELINE configs[3]'s split
                   (SURVEY section 8(e)): 5 samples per GPU at N = 8.

Synthetic data: real 3dpf geometry + random categorical features / ESM block, random-init weights (no network for
checkpoints).  Inputs are resident in HBM before the timed region.

The JSON line also carries
  roofline      the dominant kernel (ddp_conv_rows16_kernel since round 6, ddp_conv_rows_kernel in round 5; matrix-core bound by its instruction mix).  `achieved` =
                matrix-core instruction FLOPs of the kernel's own formulation without padding - the two fc products and (rows kernel) the
                per-edge G contraction as fp16 hi/lo split products = three fp16 MFMA FLOPs per product FLOP, the rest fp32 - / mean
                launch time from HIP events on the launch stream over the instrumented steps behind the timed region; `peak` = the same
                FLOPs / the time they take at each instruction's dense peak (2500 fp16, 157.3 fp32 TFLOP/s); `frac` = achieved / peak.
                `l2` carries the counter-derived L2 -> CU traffic of the kernel against the rate the guide measures for rows shared
                through the XCD L2s.  The reference formulation's FLOPs, most of which the source-node
                factorisation removes, are reported separately as `algorithmic_vs_fp32_peak`.  PMC-derived fields come from
                profiles/r06_pmc.json (collected on the builder's box, see `pmc_source`) and are dropped when that file was taken from
                other kernel sources than the ones loaded (source hash).
  cpu_baseline  the CPU oracle (reference-equivalent restatement, kind "port") on a bounded sample of the same workload
  other_workloads  BASELINE configs[2] (flexible side chains) and configs[0] (cfg1, 4 samples) measured in the same run.
"""
import argparse
import functools
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
F16_MFMA_PEAK_TFLOPS = 2500.0   # same guide: dense fp16 / bf16 MFMA (v_mfma_f32_32x32x16_f16)


def mixed_roofline(fc16, other, sec):
    """Roofline of a conv launch whose fc products (fc16 FLOPs) run as fp16 hi/lo split products - THREE fp16 matrix-core FLOPs per
    product FLOP - and whose other useful FLOPs (G pass, contraction; fp32-MFMA launches) run in fp32: the time both parts would
    take at their instruction's dense peak against the measured time.  Returns (achieved TFLOP/s of ISSUED-useful instruction
    FLOPs, the blended peak of this instruction mix, frac = achieved / peak = t_at_peak / t)."""
    issued = 3.0 * fc16 + other
    t_min = 3.0 * fc16 / (F16_MFMA_PEAK_TFLOPS * 1e12) + other / (FP32_MFMA_PEAK_TFLOPS * 1e12)
    return issued / sec / 1e12, issued / t_min / 1e12, t_min / sec
HBM_PEAK_GBS = 8000.0           # same guide: HBM3E ~8 TB/s
SUSTAINED_F16_MFMA_TFLOPS = 1700.0     # measured: every SIMD issuing v_mfma_f32_32x32x16_f16 back to back (1.6 - 1.75 PFLOP/s at 1.55 - 1.75 GHz)
SUSTAINED_TILE_LOOP_TFLOPS = 1160.0    # measured: the stream-tile loop of ddp_conv_rows alone (388 TFLOP/s fp32-equivalent x 3)
PMC_FILE = os.environ.get("DDP_PMC_FILE", os.path.join("profiles", "r06_pmc.json"))     # (DDP_PMC_FILE: a counter file of this very run, tools/gpu_round.sh)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=40)
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="auto: one rank -> the 40-sample job; several ranks -> the STRONG split of the 40 samples is the headline "
                         "(BASELINE configs[3]) and the weak run (40 samples per rank) is measured in the same invocation as a sub-record")
    ap.add_argument("--cfg", default="cfg2", choices=["cfg1", "cfg2", "small32"])
    ap.add_argument("--flex", action="store_true", help="flexible side chains (BASELINE configs[2])")
    ap.add_argument("--ways", type=int, default=1,
                    help="resident sample groups per GPU (sampler.PipelinedSampler); measured slower than one batch on MI355X "
                         "(DESIGN.md section 4.5), hence 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-pass", action="store_true",
                    help="skip the two extra steps that time the HBM-bound kernels (use under rocprofv3 so that its per-kernel "
                         "means cover warm-up + timed steps only)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the configs[2] / configs[0] sub-records")
    ap.add_argument("--launch-log", default=None,
                    help="write the per-launch (kernel, edges, useful FLOPs) list of every conv launch to this JSON file "
                         "(tools/pmc_collect.py matches it with the PMC dispatches of the same run)")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="launcher test (no GPU needed): every rank prints its RANK / WORLD_SIZE / sample slice as one JSON line and "
                         "exits, rank DDP_BENCH_FAIL_RANK (if set) with code 3")
    ap.add_argument("--cpu-samples", type=int, default=40, help="sample graphs per CPU-baseline step (40 = the full workload)")
    ap.add_argument("--cpu-batch", type=int, default=4, help="graphs per oracle forward (memory: ~1 GB of per-edge weights per graph and conv)")
    ap.add_argument("--cpu-budget-s", type=float, default=60.0, help="wall-clock budget of the CPU baseline (a bounded sample)")
    ap.add_argument("--cpu-full", action="store_true", help="no budget: two FULL 40-sample steps and three full cfg1 loops")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="threads of the CPU baseline (capped at os.cpu_count(); 0 = all cores - on the GPU boxes' many-core hosts the "
                         "oracle's small ops get SLOWER beyond ~32 threads)")
    ap.add_argument("--no-roofline-pass", action="store_true", help="skip the instrumented steps behind the timed region")
    ap.add_argument("--no-fork-front", action="store_true",
                    help="diagnostic: the front's chains and the index lists one after the other (model.fork_front = False) for same-box A/B runs")
    ap.add_argument("--concurrent-max-atoms", type=int, default=None,
                    help="diagnostic: model.concurrent_max_atoms (batches with more pocket atoms take the large-batch launch order) for A/B runs")
    ap.add_argument("--no-fork-lists-flex", action="store_true", help="diagnostic: model.fork_lists_flex = False (A/B runs)")
    ap.add_argument("--no-fork-means", action="store_true", help="diagnostic: model.fork_small_means = False (A/B runs)")
    ap.add_argument("--no-split-rows", action="store_true",
                    help="diagnostic: the factorised convs of a layer as ONE launch behind all of stage A (model.split_rows_launch = False) "
                         "for same-box A/B runs")
    ap.add_argument("--no-overlap-direct", action="store_true",
                    help="diagnostic: the serial launch order of the conv layers (model.overlap_direct_conv = False) for same-box A/B runs")
    ap.add_argument("--no-flex-sharing", action="store_true",
                    help="diagnostic: flexible side chains without the partial sharing of layers 0 / 1 (model.share_flex_layer0 = False)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------- rank launcher
def spawn_ranks(args, argv):
    """Parent of an N-rank run: N fresh children, one per GPU.  Runs before anything in this process touches the GPU (no
    torch.cuda call, no HIP library loaded); the children are new interpreters, never an exec of this one."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    # poll: as soon as one rank fails the others are terminated (they would otherwise sit in RCCL init / the final
    # all_gather until a timeout)
    rc, live = 0, dict(enumerate(procs))
    while live:
        for r, p in list(live.items()):
            code = p.poll()
            if code is None:
                continue
            del live[r]
            if code != 0:
                print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
                if rc == 0:
                    rc = code or 1
                    for q in live.values():
                        q.terminate()
        if live:
            time.sleep(0.05)
    return rc


def source_hash():
    from diffdock_pocket_amd.build import source_hash as h
    return h()


def loaded_hash():
    """ddp_source_hash() of the library that is actually loaded (a DDP_HIP_LIB override included)."""
    from diffdock_pocket_amd import _lib
    return _lib.load().ddp_source_hash().decode()


def load_pmc(workload_key):
    """PMC-derived per-launch figures (tools/pmc_collect.py -> PMC_FILE).  Returned only if they were collected
    on the kernel sources that are loaded now AND on this workload; otherwise {} (the fields are then null in the line)."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            pmc = json.load(f)
    except OSError:
        return {}, None
    src = {"file": PMC_FILE, "src_sha16": pmc.get("src_sha16"), "workload": pmc.get("workload"),
           "measured_on": pmc.get("measured_on", "builder's gpurun box (one MI355X), rocprofv3 --pmc passes of tools/gpu_round.sh; NOT this run")}
    if pmc.get("src_sha16") != loaded_hash() or pmc.get("workload") != workload_key:
        src["stale"] = True
        return {}, src
    kernels = dict(pmc.get("kernels", {}))
    kernels["_per_step"] = pmc.get("_per_step")
    return kernels, src


def model_kwargs(cfg, flex):
    if cfg == "cfg2":          # the README's large score model (reference README.md:72)
        ns, nv, L, emb = 60, 10, 6, 64
    elif cfg == "small32":     # the README's small score model AS THE README DEFINES IT (reference README.md:82: --ns 32 --nv 6
        ns, nv, L, emb = 32, 6, 5, 32   # --num_conv_layers 5 --atom_max_neighbors 12 --tr_sigma_max 15, embedding widths at the parser's 32)
    else:                      # cfg1 = BASELINE configs[0]
        ns, nv, L, emb = 16, 4, 2, 32
    return dict(sh_lmax=1, ns=ns, nv=nv, num_conv_layers=L, sigma_embed_dim=emb, distance_embed_dim=emb,
                cross_distance_embed_dim=emb, lig_max_radius=5.0, cross_max_distance=80.0, dynamic_max_cross=True,
                scale_by_sigma=True, batch_norm=True, dropout=0.0, lm_embedding_type="esm", fixed_center_conv=True,
                atom_max_neighbors=12 if cfg == "small32" else 8, flexible_sidechains=flex, use_old_atom_encoder=False), emb


def sigma_ranges(cfg):
    """Noise ranges the model was trained with: README.md:72 (tr_sigma_max 5) / README.md:82 (the small model: 15)."""
    from diffdock_pocket_amd.diffusion import SigmaRanges
    return SigmaRanges(tr_sigma_max=15.0) if cfg == "small32" else SigmaRanges()


def build_model(cfg, flex, device):
    import torch
    from diffdock_pocket_amd.diffusion import SigmaRanges, get_timestep_embedding, t_to_sigma
    from diffdock_pocket_amd.score_model import TensorProductScoreModel
    kw, emb = model_kwargs(cfg, flex)
    torch.manual_seed(0)
    model = TensorProductScoreModel(t_to_sigma=functools.partial(t_to_sigma, args=sigma_ranges(cfg)), device=device,
                                    timestep_emb_func=get_timestep_embedding("sinusoidal", emb, 1000.0), **kw)
    # non-trivial BatchNorm statistics (identity BN would be unrepresentative)
    g = torch.Generator().manual_seed(1)
    for name, buf in model.named_buffers():
        if name.endswith("running_var"):
            buf.copy_(torch.rand(buf.shape, generator=g) * 1.5 + 0.5)
        elif name.endswith("running_mean"):
            buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
    return model.to(device).eval(), kw


def cpu_baseline(args, model, kw, complex_graph):
    """Reference-equivalent CPU restatement (oracle/) on a BOUNDED sample of the same workload, `--cpu-threads` host cores (count
    stated).  cfg1 (BASELINE configs[0]) first: the whole 4-sample x 20-step loop three times, median (nothing extrapolated).
    cfg2: forwards of `--cpu-batch` sample graphs at t = 1.0 and t = 0.5 (first and mid schedule position), as many rounds as
    the rest of `--cpu-budget-s` allows or until the 40 graphs of the step are done (`--cpu-full`: always all 40: SURVEY section
    8(d)'s two full steps, ~10 minutes on the GPU box's host; profiles/r03_cpu_baseline_full.json is such a run); the graphs
    timed (`extrapolated_from_graphs`) and the per-batch times are listed - they are what "linear in the graph count" rests on."""
    import statistics
    import numpy as np
    import torch
    from oracle.ref_model import OracleConfig, OracleScoreModel
    from diffdock_pocket_amd.batch import collate, set_time
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    n, bs = args.cpu_samples, max(1, args.cpu_batch)
    budget = float("inf") if args.cpu_full else float(args.cpu_budget_s)
    torch.set_num_threads(min(args.cpu_threads or (os.cpu_count() or 1), os.cpu_count() or 1))
    t_begin = time.perf_counter()

    def oracle_for(m, k):
        ocfg = OracleConfig(ns=k["ns"], nv=k["nv"], num_conv_layers=k["num_conv_layers"], sigma_embed_dim=k["sigma_embed_dim"],
                            distance_embed_dim=k["distance_embed_dim"], cross_distance_embed_dim=k["cross_distance_embed_dim"],
                            flexible_sidechains=k["flexible_sidechains"], embedding_scale=1000.0)
        return OracleScoreModel(ocfg, {kk: v.detach().cpu() for kk, v in m.state_dict().items()})

    # BASELINE configs[0] first: cfg1, 4 samples x 20 steps, the WHOLE sampling loop on the CPU, three times (median; ~3.5 s each)
    m1, kw1 = build_model("cfg1", True, torch.device("cpu"))
    o1 = oracle_for(m1, kw1)
    g1 = make_3dpf_complex(seed=0, flexible_sidechains=True)
    sched = get_t_schedule(20)
    runs = []
    for rep in range(3):
        smp = Sampler(lambda bb: o1(bb), g1, 4, torch.device("cpu"), SamplerConfig(inference_steps=20, flexible_sidechains=True), seed=0)
        smp.randomize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(20):
                smp.step(i, sched)
        runs.append(time.perf_counter() - t0)
    cfg1 = {"value": 4.0 / statistics.median(runs), "unit": "poses/s", "seconds_per_20_step_loop": runs, "cores": torch.get_num_threads(),
            "note": "median of 3 full 20-step loops (nothing extrapolated)"}

    # the headline workload: as many `--cpu-batch`-graph forwards as the remaining budget allows, alternating over the steps
    oracle = oracle_for(model, kw)
    gs = []
    g = torch.Generator().manual_seed(7)
    for _ in range(n):
        c = complex_graph.clone()
        c["ligand"].pos = c["ligand"].pos + torch.randn(1, 3, generator=g) * 2.0
        gs.append(c)
    with torch.no_grad():     # warm-up: one single-graph forward (thread pool, allocator)
        b = collate(gs[:1])
        set_time(b, 1.0, 1.0, 1.0, 1.0)
        oracle(b)
    steps_t = (1.0, 0.5)[: args.cpu_steps]
    batches = [[] for _ in steps_t]
    last = None
    for i in range(0, n, bs):
        for si, t in enumerate(steps_t):
            # one more BATCH while it fits the budget (every step position is timed at least once)
            if last is not None and all(batches) and (time.perf_counter() - t_begin) + last > budget:
                break
            t0 = time.perf_counter()
            b = collate(gs[i:i + bs])
            set_time(b, t, t, t, t)
            with torch.no_grad():
                oracle(b)
            last = time.perf_counter() - t0
            batches[si].append((len(gs[i:i + bs]), last))
        else:
            continue
        break
    per_graph = [sum(x[1] for x in bt) / sum(x[0] for x in bt) for bt in batches]
    s_per_step = float(np.mean(per_graph)) * n
    n_timed = [sum(x[0] for x in bt) for bt in batches]
    out = {"value": n / (s_per_step * 20.0), "unit": "poses/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"oracle/ref_model.py (fp32 PyTorch-CPU, per-edge weights materialised) on {bs}-graph batches of the workload's {n} "
                     f"sample graphs at t = {', '.join(str(t) for t in steps_t)}: {', '.join(str(k) for k in n_timed)} "
                     f"graphs timed per step ({', '.join(f'{p:.2f}' for p in per_graph)} s per graph), scaled to {n} graphs x 20 steps",
           "extrapolated_from_graphs": n_timed, "graphs_per_step_of_the_workload": n,
           "seconds_per_batch": [[round(x[1], 3) for x in bt] for bt in batches], "graphs_per_batch": bs,
           "seconds_per_40_sample_step": s_per_step}
    out["configs[0] cfg1 4 samples x 20 steps (whole CPU loop)"] = cfg1
    out["seconds_spent"] = time.perf_counter() - t_begin
    return out


def timed_job(model, complex_graph, n_total, sl, device, flex, steps, warmup, ways=1, sync=lambda: None, dist=None, world=1,
              on_timed=None, cfg="cfg2"):
    """`warmup` (at least 3) untimed steps on the sampler that is then timed - they fill the model's static caches and the
    allocator and capture the step's hipGraph - then the job restarts from its first step (poses and noise stream restored)
    and exactly `steps` steps are timed, bracketed by barrier + synchronize.  on_timed(sampler, snapshot, schedule) runs after
    the timed region (the instrumented roofline pass).  Returns (seconds, sampler, final poses, gathered poses, schedule, info)."""
    import torch
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import PipelinedSampler, Sampler, SamplerConfig
    scfg = SamplerConfig(inference_steps=20, flexible_sidechains=flex, sigma=sigma_ranges(cfg))
    if ways > 1:
        sampler = PipelinedSampler(model, complex_graph, n_total, device, scfg, seed=0, sample_slice=sl, ways=ways)
    else:
        sampler = Sampler(model, complex_graph, n_total, device, scfg, seed=0, sample_slice=sl)
    sampler.randomize()
    schedule = get_t_schedule(20)
    snap = sampler.snapshot() if ways == 1 else None
    for i in range(max(warmup, 3)):
        sampler.step((i * 10) % 20, schedule)   # schedule positions 0, 10, ..: the largest edge sets (allocator) and a typical step
    if snap is not None:
        sampler.restore(snap)     # the timed steps see the schedule's own poses and edge counts
    else:
        sampler = PipelinedSampler(model, complex_graph, n_total, device, scfg, seed=0, sample_slice=sl, ways=ways)
        sampler.randomize()
    torch.cuda.synchronize()
    sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        sampler.step(i % 20, schedule)
    final_pos = sampler.lig_pos.contiguous()
    gathered, info = None, {"hip_graph": bool(getattr(sampler, "_graph", None))}
    if dist is not None:   # gather final ligand poses of all shards (RCCL over xGMI); shards may differ in size (strong scaling)
        sizes = [len(range(*shard_slice(r, world, n_total, None).indices(n_total))) for r in range(world)]
        pad = max(sizes)
        buf = torch.zeros((pad,) + tuple(final_pos.shape[1:]), device=device, dtype=final_pos.dtype)
        buf[: final_pos.shape[0]] = final_pos
        out = [torch.empty_like(buf) for _ in range(world)]
        torch.cuda.synchronize()
        tg = time.perf_counter()
        dist.all_gather(out, buf)
        torch.cuda.synchronize()
        info["all_gather_ms"] = (time.perf_counter() - tg) * 1e3     # (includes waiting for the slowest rank's last step)
        gathered = torch.cat([o[:s] for o, s in zip(out, sizes)], 0)
    torch.cuda.synchronize()
    sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if hasattr(sampler, "check_overflow"):
        sampler.check_overflow()      # replayed steps never enter the model's Python forward: a truncated edge list is reported here
    info["edges_last_step"] = dict(getattr(sampler, "last_stats", None) or model.last_stats)     # (of THIS job: read before any other runs)
    if on_timed is not None:
        on_timed(sampler, snap, schedule)
    return elapsed, sampler, final_pos.clone(), gathered, schedule, info


def shard_hbm_plan(cfg, n_local):
    """HBM a rank's shard of `n_local` sample graphs needs for the two big per-layer arrays of the forward - sized PER SHARD, never
    per job: G (stage A's output, read once by the 32-edge conv kernel: one row of DDP_G_LD floats per source node, conv and G slot)
    and the messages.  Host arithmetic only (packing specs + the 3dpf geometry file): no GPU, no model."""
    import numpy as np
    from diffdock_pocket_amd import packing as P
    kw, _ = model_kwargs(cfg, False)
    ns, nv, L = kw["ns"], kw["nv"], kw["num_conv_layers"]
    with np.load(os.path.join(ROOT, "diffdock_pocket_amd", "assets", "3dpf_geometry.npz")) as z:
        n_l, n_r, n_a = int(z["lig_pos"].shape[0]), int(z["rec_pos"].shape[0]), int(z["atom_pos"].shape[0])
        e_rr = int(z["rec_edge_index"].shape[1])
    k = kw["atom_max_neighbors"]
    worst = 0
    for l in range(L):
        sg = P.faster_tp_spec(P.irreps_muls(ns, nv, l), P.irreps_muls(ns, nv, l + 1), 3 * ns, factorized=True)
        hg = (sg.hid + 3) // 4 * 4
        row = sum(((hg + 1) * gc + 31) // 32 * 32 for gc in sg.g_cols if gc > 0) * 4       # bytes per source node and conv
        g = n_local * row * (n_a + 3 * n_l + 3 * n_r)        # atom<-atom | three ligand-source convs | three receptor-source convs
        msgs = n_local * 4 * sg.d_out * (n_a * k + 2 * n_a + e_rr + 3 * n_l * 64)          # (ligand edge sets: a generous 64 per atom)
        worst = max(worst, g + msgs)
    return {"samples": n_local, "g_and_messages_bytes_largest_layer": int(worst), "hbm_bytes_per_gpu": 288 * 10 ** 9}


def shard_slice(rank, world, n_total, _):
    """Rank r of R owns samples [r*N/R, (r+1)*N/R) of the job."""
    return slice(rank * n_total // world, (rank + 1) * n_total // world)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args, argv))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with matching values "
                         f"(or without WORLD_SIZE, then bench.py starts the ranks itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    scaling = args.scaling
    if scaling == "auto":      # several ranks: the strong split of ONE complex's samples is the job BASELINE configs[3] names
        scaling = "strong" if world > 1 else "weak"
    if args.dry_run_ranks:
        n_total = args.samples * world if scaling == "weak" else args.samples
        sl = shard_slice(rank, world, n_total, None)
        # (device: what the rank WOULD bind - torch.cuda.set_device(LOCAL_RANK) below; hbm_plan: the shard's big arrays)
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "master": os.environ.get("MASTER_ADDR"),
                          "port": os.environ.get("MASTER_PORT"), "samples_total": n_total, "slice": [sl.start, sl.stop],
                          "scaling": scaling, "device": f"cuda:{local_rank}",
                          "hbm_plan": shard_hbm_plan(args.cfg, len(range(*sl.indices(n_total))))}), flush=True)
        sys.exit(3 if os.environ.get("DDP_BENCH_FAIL_RANK") == str(rank) else 0)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU product path)")
    # host-side set-up (weight packing of the models this run builds) on at most 32 threads per rank: PyTorch's small CPU ops get
    # several times slower with every core of the GPU boxes' many-core hosts (the GPU tests: 1018 -> 327 s with the same cap);
    # nothing inside a timed region runs on these threads, and cpu_baseline sets its own count (--cpu-threads)
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 1) // max(1, world))))
    test_backend = os.environ.get("DDP_BENCH_BACKEND")   # "gloo": several ranks on ONE device (1-GPU box); never a real run
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if not test_backend:
            raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but only {ndev} visible")
        local_rank %= max(ndev, 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist, backend = None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = test_backend or "nccl"
        dist.init_process_group(backend, rank=rank, world_size=world)
    if world > 1:
        sync = (lambda: dist.barrier(device_ids=[local_rank])) if backend == "nccl" else dist.barrier
    else:
        sync = lambda: None   # noqa: E731

    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    from diffdock_pocket_amd import score_model as sm

    model, kw = build_model(args.cfg, args.flex, device)
    if args.no_flex_sharing:
        model.share_flex_layer0 = False
    if args.no_overlap_direct:
        model.overlap_direct_conv = False
    if args.no_split_rows:
        model.split_rows_launch = False
    if args.no_fork_front:
        model.fork_front = False
    if args.no_fork_lists_flex:
        model.fork_lists_flex = False
    if args.no_fork_means:
        model.fork_small_means = False
    if args.concurrent_max_atoms is not None:
        model.concurrent_max_atoms = args.concurrent_max_atoms
    complex_graph = make_3dpf_complex(seed=0, flexible_sidechains=args.flex)
    n_total = args.samples * world if scaling == "weak" else args.samples
    if n_total < world:
        raise SystemExit("bench.py: fewer samples than ranks")
    sl = shard_slice(rank, world, n_total, None)
    n_local = len(range(*sl.indices(n_total)))
    ranks_seen = dist.get_world_size() if dist is not None else 1

    # Kernel-level figures (roofline): the timed region replays a captured hipGraph per step, into which no events can be
    # placed; the same `steps` steps are therefore run once more behind it, launch by launch, with HIP events around every conv
    # launch on the launch stream (prof) and, in two further steps, around the HBM-bound kernels (prof_hbm).
    prof, prof_hbm = sm.ConvProfiler(), sm.ConvProfiler()
    prof_hbm.hbm_on = True
    sm.set_conv_profiler(None)

    def roofline_pass(sampler, snap, schedule):
        if rank != 0 or args.no_roofline_pass or snap is None:
            return
        sampler.restore(snap)
        sampler.graph_enabled = False
        # (the instrumented steps launch a layer's factorised convs as ONE launch: its HIP-event time is then the kernel's, not its share of
        # the chip beside stage A of the atom rows - the timed region starts the receptor- / ligand-sourced convs early, model.split_rows_launch)
        split_was = getattr(model, "split_rows_launch", False)
        model.split_rows_launch = False
        sm.set_conv_profiler(prof)
        for i in range(args.steps):
            sampler.step(i % 20, schedule)
        torch.cuda.synchronize()
        sm.set_conv_profiler(None)
        if not args.no_hbm_pass:
            sm.set_conv_profiler(prof_hbm)
            for i in range(2):
                sampler.step(5 + 10 * i, schedule)
            torch.cuda.synchronize()
            sm.set_conv_profiler(None)
        model.split_rows_launch = split_was
        sampler.graph_enabled = True

    elapsed, sampler, final_pos, gathered, schedule, info = timed_job(model, complex_graph, n_total, sl, device, args.flex, args.steps,
                                                                      args.warmup, ways=args.ways, sync=sync, dist=dist, world=world,
                                                                      on_timed=roofline_pass, cfg=args.cfg)
    rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in allt]
        elapsed = max(float(x.item()) for x in allt)
        assert gathered.shape[0] == n_total
    assert torch.isfinite(final_pos).all(), "non-finite poses"
    weak = None
    if world > 1 and args.scaling == "auto":     # the weak figure (40 samples per rank) in the same invocation
        nw = args.samples * world
        slw = shard_slice(rank, world, nw, None)
        elw, _, fpw, gw, _, infow = timed_job(model, complex_graph, nw, slw, device, args.flex, args.steps, args.warmup, sync=sync,
                                              dist=dist, world=world, cfg=args.cfg)
        tt = torch.tensor([elw], device=device, dtype=torch.float64)
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        elw_max = max(float(x.item()) for x in allt)
        assert torch.isfinite(fpw).all() and gw.shape[0] == nw
        weak = {"scaling": "weak", "samples_total": nw, "value": nw * args.steps / 20.0 / elw_max, "unit": "poses/s",
                "ms_per_step": elw_max / args.steps * 1e3, "ms_per_step_by_rank": [float(x.item()) / args.steps * 1e3 for x in allt],
                "all_gather_ms": infow.get("all_gather_ms")}

    if rank == 0:
        workload_key = f"{args.cfg} samples={args.samples} flex={args.flex} ways={args.ways} scaling={scaling} n_gpus={world}"
        pmc, pmc_src = load_pmc(workload_key)
        poses = n_total * args.steps / 20.0
        if args.launch_log:
            prof._resolve()
            os.makedirs(os.path.dirname(os.path.abspath(args.launch_log)), exist_ok=True)
            with open(args.launch_log, "w") as f:
                json.dump({"workload": workload_key, "src_sha16": loaded_hash(), "warmup_steps": args.warmup, "steps": args.steps,
                           "launches": [{"kernel": k, "edges": e, "useful_flops": u, "algorithmic_flops": a, "fc16_flops": (f16 if h else 0.0)}
                                        for k, e, u, a, f16, h in zip(prof.kernel, prof.edges, prof.useful, prof.flops, prof.fc, prof.h2)]}, f)
        # the dominant kernel = the instantiation with the larger share of the timed region
        kinds = sorted({k for k in prof.kernel}, key=lambda k: -prof.summary(k)[2])
        roof = None
        if kinds:
            def entry(kname):
                n_, fl_, ms_ = prof.summary(kname)
                sec = ms_ / n_ * 1e-3
                useful = prof.useful_flops(kname) / n_
                issued_model = prof.executed_flops(kname) / n_
                p = pmc.get(kname, {})
                fc16, oth = prof.split_flops(kname)
                ach, peak, frac = mixed_roofline(fc16 / n_, oth / n_, sec)
                e = {"kernel": kname, "bound": "mfma", "unit": "TFLOP/s", "peak": peak, "achieved": ach, "frac": frac,
                     "mfma_mix": {"fc_products_as_fp16_hi_lo_split_gflop_per_launch": fc16 / n_ / 1e9, "fp32_gflop_per_launch": oth / n_ / 1e9,
                                  "fp16_mfma_peak": F16_MFMA_PEAK_TFLOPS, "fp32_mfma_peak": FP32_MFMA_PEAK_TFLOPS,
                                  "instruction_flops_per_split_product_flop": 3},
                     # what this chip SUSTAINS (profiles/r05_mfma_chain_micro.txt, r05_stream_wide_micro.txt): the
                     # dense peak above is the guide's 2.5 PFLOP/s at 2.4 GHz; measured on the builder's boxes, not in this run
                     "sustained_mfma": {"fp16_mfma_from_registers_tflops": SUSTAINED_F16_MFMA_TFLOPS,
                                        "tile_loop_with_lds_operands_tflops": SUSTAINED_TILE_LOOP_TFLOPS,
                                        "frac_of_mfma_from_registers": frac * F16_MFMA_PEAK_TFLOPS / SUSTAINED_F16_MFMA_TFLOPS,
                                        "frac_of_tile_loop": frac * F16_MFMA_PEAK_TFLOPS / SUSTAINED_TILE_LOOP_TFLOPS,
                                        "note": "back-to-back v_mfma_f32_32x32x16_f16 from registers on every SIMD run at 1.55 - 1.75 GHz "
                                                "(tools/micro/mfma_chain.hip); the stripped stream-tile loop of ddp_conv_rows (operands from an "
                                                "LDS ring, feature contraction) sustains 1.16 PFLOP/s of fp16 MFMA issue in the shipped "
                                                "two-wave form and at one 512-register wave per SIMD alike (tools/micro/stream_wide.hip)"},
                     "fp32_equivalent_tflops": useful / sec / 1e12, "fp32_equivalent_vs_fp32_mfma_peak": useful / sec / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                     "launches": n_, "avg_launch_ms": ms_ / n_, "useful_mfma_gflop_per_launch": useful / 1e9,
                     "tile_padded_mfma_gflop_per_launch": issued_model / 1e9,
                     "algorithmic_gflop_per_launch": fl_ / n_ / 1e9,
                     "algorithmic_vs_fp32_peak": fl_ / n_ / sec / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                     "share_of_wall": ms_ * 1e-3 / elapsed, "ms_per_step": ms_ / args.steps,
                     "by_layer": {str(tag): {"launches": n_t, "avg_launch_ms": ms_t / n_t, "edges_per_launch": ne_t / n_t,
                                             "fp32_equivalent_tflops": u_t / (ms_t * 1e-3) / 1e12,
                                             "achieved": mixed_roofline(f16_t, u_t - f16_t, ms_t * 1e-3)[0],
                                             "frac": mixed_roofline(f16_t, u_t - f16_t, ms_t * 1e-3)[2]}
                                  for tag, (n_t, u_t, ms_t, ne_t, f16_t) in sorted(prof.by_tag(kname).items(), key=lambda kv: str(kv[0]))},
                     "traffic": p.get("hbm_bytes_per_launch"), "issued_mfma_gflop_per_launch_pmc": p.get("issued_mfma_gflop_per_launch"),
                     "padding_frac_pmc": p.get("padding_frac"), "mfma_busy_pmc": p.get("mfma_busy_frac")}
                l2 = p.get("l2")
                if l2:   # counter-derived L2 -> CU traffic of the kernel (128-byte requests) against what the chip delivers from its XCD L2s
                    cyc = l2.get("cycles_per_launch") or 0.0
                    e["l2"] = {"request_bytes_per_launch": l2.get("l2_request_bytes_per_launch"),
                               # the counter passes sample the timed region's SPLIT conv launches; avg_launch_ms above is per whole-layer
                               # launch of the instrumented pass: the same bytes on that basis and per step
                               "launches_per_step_in_the_counter_pass": l2.get("launches_per_step"),
                               "request_bytes_per_step": l2.get("l2_request_bytes_per_step"),
                               "request_bytes_per_layer_launch": l2.get("l2_request_bytes_per_layer_launch"),
                               "hit_rate": l2.get("l2_hit_rate"),
                               "busy_frac": l2.get("l2_busy_frac"), "tcp_tcc_read_latency_cycles": l2.get("tcp_tcc_read_latency_cycles"),
                               "bytes_per_cycle_chip": l2.get("l2_bytes_per_cycle_chip"),
                               "tb_per_s_at_the_profiled_launch_time": (l2.get("l2_bytes_per_cycle_chip") or 0.0) * 2.1e9 / 1e12 if cyc else None,
                               "guide_measured_peak_tb_per_s": [16.8, 18.8],
                               "note": "rocprofv3 --pmc TCC_REQ_sum x 128 B per launch / GRBM_GUI_ACTIVE (per XCD) cycles; the 32-edge kernel of round 4 "
                                       "read 15.3 TB/s here with the L2 90 % busy (profiles/r05_pmc_conv32_h2_l1_l2.json)"}
                return e
            roof = entry(kinds[0])
            roof["note"] = ("achieved = matrix-core instruction FLOPs of the kernel's own formulation without padding (fc1, the vector-feature fc2 "
                            "columns and - ddp_conv_rows - the per-edge G contraction as fp16 hi/lo split products = 3 fp16 MFMA FLOPs per "
                            "product FLOP; the feature contraction in fp32) / HIP-event launch time; peak = the same FLOPs / the time they "
                            "take at each instruction's dense peak (2500 fp16, 157.3 fp32 TFLOP/s), so frac = time at peak / measured time; "
                            "ddp_conv_rows runs a whole 32 x 32 tile product per run of edges with one source node, of which only the run's "
                            "rows are useful (counted) - the issued work is in issued_mfma_gflop_per_launch_pmc; "
                            "fp32_equivalent_* counts every product FLOP once; algorithmic_* = the reference formulation "
                            "(BASELINE.md section 3: 2FH + 2HW + 2C per edge), 84 % of which the exact source-node factorisation "
                            "removes (DESIGN.md section 4.2), hence algorithmic_vs_fp32_peak > 1")
            roof["conv_share_of_wall"] = sum(prof.summary(k)[2] for k in kinds) * 1e-3 / elapsed
            roof["pmc_source"] = pmc_src
            roof["other_kernels"] = [entry(k) for k in kinds[1:]]

            def hbm_entry(kname):   # HBM-bound kernels of the path: algorithmic bytes / HIP-event time against ~8 TB/s
                n_, by_, ms_ = prof_hbm.hbm_summary(kname)
                if n_ == 0:
                    return None
                gbs = by_ / (ms_ * 1e-3) / 1e9
                p = pmc.get(kname, {})
                return {"kernel": kname, "bound": "hbm", "launches": n_, "avg_launch_ms": ms_ / n_, "achieved": gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_mb_per_launch": by_ / n_ / 1e6,
                        "ms_per_step": ms_ / 2.0, "measured": "2 extra steps (schedule positions 5, 15) behind the instrumented pass; launches of "
                        "the layer order's parallel chains overlap: a launch's time includes the share of the chip it left to the others "
                        "(standalone rates: tools/bench_stage_a.py, DESIGN.md section 4.8), and ms_per_step sums overlapping launches",
                        "traffic": p.get("hbm_bytes_per_launch"), "mfma_busy_pmc": p.get("mfma_busy_frac")}
            roof["other_kernels"] += [e for e in (hbm_entry("ddp_stage_a_h2_kernel"), hbm_entry("ddp_stage_a_mfma_kernel"),
                                                  hbm_entry("ddp_segment_reduce4_kernel")) if e]
            # whole step: HBM bytes the PMC passes saw against the algorithmic boundary bytes of the convs
            alg_step = prof.boundary_bytes() / args.steps
            step = {"algorithmic_boundary_gb": alg_step / 1e9}
            per_step_pmc = pmc.get("_per_step")
            if per_step_pmc:
                step["traffic_gb_pmc"] = per_step_pmc["hbm_bytes_per_step"] / 1e9
                step["traffic_vs_algorithmic"] = per_step_pmc["hbm_bytes_per_step"] / alg_step
            roof["per_step"] = step
        line = {"metric": "ligand poses/sec (40 samples x 20 steps) on 3dpf", "value": poses / elapsed, "unit": "poses/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": scaling if world > 1 else None, "vs_baseline": None,
                "dtype": "f32 (fc and G products: fp16 hi/lo split of both operands on v_mfma_f32_16x16x32_f16 - the row-stationary conv kernel - and v_mfma_f32_32x32x16_f16, fp32 accumulate; G between its two kernels as fp16 hi + continuation byte = 19 significant bits; the exact fp32 MFMA form is timed in other_workloads)", "data": "synthetic",
                "config": {"workload": f"3dpf ({sampler.n_l} lig atoms, 139 residues, {sampler.n_a} pocket atoms), "
                                       f"{n_total} samples over {world} GPU(s) ({n_local} on rank 0) x 20-step schedule, score model "
                                       f"{args.cfg} (ns={kw['ns']} nv={kw['nv']} L={kw['num_conv_layers']}), "
                                       f"flexible_sidechains={args.flex}", "samples_total": n_total, "samples_rank0": n_local,
                           "parallelism": f"samples sharded over {world} rank(s), one final all_gather of poses" +
                                          ("; NO run on more than one GPU exists (no multi-GPU node was available to the builder)" if world == 1 else ""),
                           "ways": args.ways, "ms_per_step_by_rank": rank_ms, "src_sha16": loaded_hash(), "device": str(device),
                           "hbm_plan_rank0": shard_hbm_plan(args.cfg, n_local),
                           "hip_graph_replay": info.get("hip_graph"), "rccl_ranks_seen": ranks_seen, "backend": backend,
                           "all_gather_ms": info.get("all_gather_ms"),
                           "edges_last_step": info.get("edges_last_step")},
                "roofline": roof}
        if weak is not None:
            line["weak_scaling"] = weak
        if roof is not None:
            roof["measured"] = (f"{args.steps} instrumented steps behind the timed region (same poses, noise and schedule positions, "
                                "launch by launch with HIP events on the launch stream); the timed region itself replays one captured "
                                "hipGraph per step")
        default_workload = args.samples == 40 and args.cfg == "cfg2" and not args.flex and args.ways == 1
        if world == 1 and default_workload and not args.no_other_workloads:
            # BASELINE configs[2] and configs[0] in the same driver-run line (short runs: 20 and 20 steps)
            others = {}

            def conv_fracs(pr):   # per conv kernel: useful fp32-MFMA TFLOP/s of the instrumented pass against the peak
                out = {}
                for kname in sorted({k for k in pr.kernel}):
                    n_, _, ms_ = pr.summary(kname)
                    tf = pr.useful_flops(kname) / (ms_ * 1e-3) / 1e12
                    fc16, oth = pr.split_flops(kname)
                    ach, peak, frac = mixed_roofline(fc16, oth, ms_ * 1e-3)
                    out[kname] = {"launches": n_, "avg_launch_ms": ms_ / n_, "fp32_equivalent_tflops": tf, "achieved_tflops": ach,
                                  "peak_of_the_instruction_mix": peak, "frac": frac}
                return out

            if hasattr(sampler, "close"):
                sampler.close()     # the main job's captured step goes back to the shared graph memory pool before the sub-records

            def sub_job(cfg_, flex_, n_, conv_h2=True, conv_rows=True, rows16=None):   # a timed job + its instrumented pass
                m_, kw_ = build_model(cfg_, flex_, device)
                m_.fork_front = model.fork_front
                m_.conv_h2 = conv_h2
                if rows16 is not None:
                    m_.rows_mfma16 = rows16
                from diffdock_pocket_amd import launch as launch_
                rows_was = launch_.CONV_ROWS
                launch_.CONV_ROWS = rows_was and conv_rows
                g_ = make_3dpf_complex(seed=0, flexible_sidechains=flex_)
                pr = sm.ConvProfiler()

                def inst(smp_, snap_, sched_):
                    smp_.restore(snap_)
                    smp_.graph_enabled = False
                    split_was_ = getattr(m_, "split_rows_launch", False)
                    m_.split_rows_launch = False      # (as in roofline_pass)
                    sm.set_conv_profiler(pr)
                    try:
                        for i in range(20):
                            smp_.step(i, sched_)
                        torch.cuda.synchronize()
                    finally:
                        sm.set_conv_profiler(None)
                        m_.split_rows_launch = split_was_
                        smp_.graph_enabled = True

                try:      # (the module-level switch is restored whatever a sub-record does: the records behind it must not inherit it)
                    el_, s_, fp_, _, _, _ = timed_job(m_, g_, n_, slice(0, n_), device, flex_, 20, 3, on_timed=inst, cfg=cfg_)
                    assert torch.isfinite(fp_).all() and torch.isfinite(s_.atom_pos).all()
                    s_.close()
                finally:
                    launch_.CONV_ROWS = rows_was
                return {"value": n_ / el_, "unit": "poses/s", "ms_per_step": el_ / 20 * 1e3, "steps": 20,
                        "edges_last_step": dict(m_.last_stats), "conv_kernels": conv_fracs(pr)}

            # the headline workload in the exact fp32 MFMA form of every fc product (model.conv_h2 = False: the kernels of rounds 1 - 3), and
            # in the h2 form through the 32-edge kernel of round 4 (launch.CONV_ROWS = False): driver-timed beside the headline
            others["configs[1] exact fp32 MFMA form (model.conv_h2 = False), 40 samples, cfg2, rigid"] = sub_job("cfg2", False, 40, conv_h2=False)
            others["configs[1] fp16 hi/lo form through the 32-edge kernel of round 4 (launch.CONV_ROWS = False)"] = sub_job("cfg2", False, 40, conv_rows=False)
            others["configs[1] through the row-stationary kernel of round 5 on v_mfma_f32_32x32x16_f16 (model.rows_mfma16 = False)"] = \
                sub_job("cfg2", False, 40, rows16=False)
            others["configs[2] 3dpf flexible side chains, 40 samples, cfg2"] = sub_job("cfg2", True, 40)
            others["configs[0] 3dpf 4 samples, cfg1 (ns=16 nv=4 L=2), flexible side chains"] = sub_job("cfg1", True, 4)
            others["README small score model (README.md:82: ns=32 nv=6 L=5, atom_max_neighbors=12, tr_sigma_max=15), 40 samples, rigid receptor"] = \
                sub_job("small32", False, 40)
            # BASELINE configs[3]'s shard: 5 of the 40 samples on this GPU (what one of 8 ranks runs under the strong split)
            el5, s5, fp5, _, _, _ = timed_job(model, complex_graph, 40, slice(0, 5), device, False, 20, 3, cfg=args.cfg)
            others["configs[3] shard: samples [0, 5) of the 40 on one GPU, cfg2"] = {
                "value": 5.0 / el5, "unit": "poses/s per GPU", "ms_per_step": el5 / 20 * 1e3, "steps": 20}
            s5.close()
            del s5
            line["other_workloads"] = others
            # the figures a strict reader of `dtype` / `config` wants, as scalars inside `config` (the driver's record keeps `config` whole and
            # only the tail of the rest): the exact-fp32 form of the headline, configs[2], configs[0], the README model, configs[3]'s shard
            def pick(prefix):
                for k_, v_ in others.items():
                    if k_.startswith(prefix):
                        return {"value": v_["value"], "ms_per_step": v_["ms_per_step"]}
                return None
            line["config"]["also_measured"] = {
                "unit": "poses/s (per GPU for the shard)",
                "exact_fp32_mfma_form": pick("configs[1] exact fp32"),
                "round4_32_edge_kernel": pick("configs[1] fp16 hi/lo form through the 32-edge"),
                "round5_rows_kernel_32x32x16": pick("configs[1] through the row-stationary kernel of round 5"),
                "configs2_flexible_side_chains": pick("configs[2]"),
                "configs0_cfg1_4_samples": pick("configs[0]"),
                "readme_small_model": pick("README small"),
                "configs3_shard_5_of_40_samples": pick("configs[3] shard")}
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only (the other ranks must not wait for it)
            line["cpu_baseline"] = cpu_baseline(args, model, kw, complex_graph)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
