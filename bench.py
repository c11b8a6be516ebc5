#!/usr/bin/env python3
"""bench.py - ligand poses/s of the reverse-diffusion hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--samples 40] [--flex] [--cfg cfg2|cfg1]

One "step" = one denoising step over the batch of `--samples` sample graphs of the 3dpf complex: graph construction
+ score-model forward (HIP) + SDE pose update.  Default workload = BASELINE.json configs[1]: 3dpf, 40 samples, full
score model (ns=60 nv=10, 6 conv layers), 20-step schedule; K steps walk the 20-step schedule cyclically, so the
default K=20 is exactly one 40-sample x 20-step job.  poses/s = samples * K / 20 / seconds  (one pose = one sample
carried through 20 denoising steps), aggregated over ranks (weak scaling: every rank runs its own 40 samples; the only
collective is the RCCL all_gather of final ligand poses at the end of the timed region).

Synthetic data: real 3dpf geometry + random categorical features / ESM block, random-init weights (no network for
checkpoints).  Inputs are resident in HBM before the timed region.

The JSON line also carries
  roofline:      the fused conv kernel (ddp_conv_messages_kernel), fp32-MFMA bound: algorithmic FLOPs per launch
                 (BASELINE.md §3 formula x actual edge counts) / mean launch time from HIP events in the timed region
  cpu_baseline:  the CPU oracle (reference-equivalent restatement, kind "port") on the same workload, bounded sample.
"""
import argparse
import functools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
HBM_PEAK_GBS = 8000.0           # same guide: HBM3E ~8 TB/s


def _load_traffic():
    """HBM bytes per conv launch, per kernel instantiation, from the rocprofv3 PMC passes (profiles/*_traffic.json,
    written by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs with the gfx950 corrections); None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            return {k: v["hbm_bytes_per_launch"] for k, v in json.load(f).get("kernels", {}).items()}
    except OSError:
        return None


TRAFFIC_BYTES_PER_LAUNCH = None


def _load_mfma_busy():
    """Matrix-pipe busy fraction per kernel from the rocprofv3 PMC pass (profiles/r01_mfma_pmc.json, tools/pmc_mfma.py:
    SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE x SIMDs); {} if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_mfma_pmc.json")) as f:
            return {k: v.get("mfma_busy_frac") for k, v in json.load(f).get("kernels", {}).items()}
    except OSError:
        return {}


def model_kwargs(cfg, flex):
    if cfg == "cfg2":
        ns, nv, L, emb = 60, 10, 6, 64
    else:
        ns, nv, L, emb = 16, 4, 2, 32
    return dict(sh_lmax=1, ns=ns, nv=nv, num_conv_layers=L, sigma_embed_dim=emb, distance_embed_dim=emb,
                cross_distance_embed_dim=emb, lig_max_radius=5.0, cross_max_distance=80.0, dynamic_max_cross=True,
                scale_by_sigma=True, batch_norm=True, dropout=0.0, lm_embedding_type="esm", fixed_center_conv=True,
                atom_max_neighbors=8, flexible_sidechains=flex, use_old_atom_encoder=False), emb


def build_model(cfg, flex, device):
    from diffdock_pocket_amd.diffusion import SigmaRanges, get_timestep_embedding, t_to_sigma
    from diffdock_pocket_amd.score_model import TensorProductScoreModel
    kw, emb = model_kwargs(cfg, flex)
    torch.manual_seed(0)
    model = TensorProductScoreModel(t_to_sigma=functools.partial(t_to_sigma, args=SigmaRanges()), device=device,
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
    """Reference-equivalent CPU restatement (oracle/) on a bounded sample of the same workload."""
    from oracle.ref_model import OracleConfig, OracleScoreModel
    from diffdock_pocket_amd.batch import collate, set_time
    n = args.cpu_samples
    ocfg = OracleConfig(ns=kw["ns"], nv=kw["nv"], num_conv_layers=kw["num_conv_layers"],
                        sigma_embed_dim=kw["sigma_embed_dim"], distance_embed_dim=kw["distance_embed_dim"],
                        cross_distance_embed_dim=kw["cross_distance_embed_dim"],
                        flexible_sidechains=kw["flexible_sidechains"], embedding_scale=1000.0)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    oracle = OracleScoreModel(ocfg, sd)
    torch.set_num_threads(min(os.cpu_count() or 1, args.cpu_threads))
    gs = []
    g = torch.Generator().manual_seed(7)
    for _ in range(n):
        c = complex_graph.clone()
        c["ligand"].pos = c["ligand"].pos + torch.randn(1, 3, generator=g) * 2.0
        gs.append(c)
    times = []
    for t in (1.0, 0.5)[: args.cpu_steps]:
        b = collate(gs)
        set_time(b, t, t, t, t)
        t0 = time.perf_counter()
        with torch.no_grad():
            oracle(b)
        times.append(time.perf_counter() - t0)
    s_per_step = float(np.mean(times))
    return {"value": n / (s_per_step * 20.0), "unit": "poses/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} denoising step(s) (t=1.0,0.5) of {n} sample graph(s) of the same workload through "
                      f"oracle/ref_model.py (fp32 PyTorch-CPU, per-edge weights materialised); {s_per_step:.2f} s/step, "
                      f"extrapolated to 20 steps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=40)
    ap.add_argument("--cfg", default="cfg2", choices=["cfg1", "cfg2"])
    ap.add_argument("--flex", action="store_true", help="flexible side chains (BASELINE configs[2])")
    ap.add_argument("--ways", type=int, default=1,
                    help="resident sample groups per GPU (sampler.PipelinedSampler: groups stepped alternately, a group's front on a "
                         "high-priority stream beside the other group's conv layers).  Measured slower than one batch on MI355X "
                         "(DESIGN.md section 4.5), hence 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-pass", action="store_true",
                    help="skip the two extra steps that time the HBM-bound kernels (use under rocprofv3 so that its per-kernel "
                         "means cover warm-up + timed steps only)")
    ap.add_argument("--cpu-samples", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=32)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU product path)")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if os.environ.get("DDP_BENCH_BACKEND") else local_rank
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") is the backend of every real run; DDP_BENCH_BACKEND=gloo only exists to exercise the multi-rank code
        # path with several ranks on ONE device (tests on a 1-GPU box), where RCCL refuses duplicate devices
        backend = os.environ.get("DDP_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world)
    if world > 1:
        sync = (lambda: dist.barrier(device_ids=[local_rank])) if backend == "nccl" else dist.barrier
    else:
        sync = lambda: None   # noqa: E731

    from diffdock_pocket_amd.sampler import PipelinedSampler, Sampler, SamplerConfig
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    from diffdock_pocket_amd import score_model as sm

    model, kw = build_model(args.cfg, args.flex, device)
    complex_graph = make_3dpf_complex(seed=0, flexible_sidechains=args.flex)
    scfg = SamplerConfig(inference_steps=20, flexible_sidechains=args.flex)
    # weak scaling: every rank owns `samples` samples of the job of world*samples samples
    n_total = args.samples * world
    def make_sampler():
        sl = slice(rank * args.samples, (rank + 1) * args.samples)
        if args.ways > 1:
            return PipelinedSampler(model, complex_graph, n_total, device, scfg, seed=0, sample_slice=sl, ways=args.ways)
        return Sampler(model, complex_graph, n_total, device, scfg, seed=0, sample_slice=sl)

    sampler = make_sampler()
    sampler.randomize()
    schedule = get_t_schedule(20)

    def one_step(i):
        sampler.step(i % 20, schedule)

    for i in range(args.warmup):
        one_step((i * 10) % 20)   # schedule positions 0, 10, ..: the largest edge sets (allocator) and a typical step
    # restart from fresh poses so that the timed steps see the schedule's own edge counts
    sampler = make_sampler()
    sampler.randomize()

    prof = sm.ConvProfiler()
    sm.set_conv_profiler(prof)
    torch.cuda.synchronize()
    sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    final_pos = sampler.lig_pos.contiguous()
    if dist is not None:   # gather final ligand poses of all shards (RCCL over xGMI)
        out = [torch.empty_like(final_pos) for _ in range(world)]
        dist.all_gather(out, final_pos)
    torch.cuda.synchronize()
    sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sm.set_conv_profiler(None)
    # the HBM-bound kernels (stage A, segment reduce) are timed in two extra steps AFTER the timed region: ~45 more event
    # pairs per step would otherwise sit inside it (measured: +4 % on ms_per_step)
    prof_hbm = sm.ConvProfiler()
    prof_hbm.hbm_on = True
    if rank == 0 and not args.no_hbm_pass:
        sm.set_conv_profiler(prof_hbm)
        for i in range(2):
            one_step(5 + 10 * i)
        torch.cuda.synchronize()
        sm.set_conv_profiler(None)
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(final_pos).all(), "non-finite poses"

    if rank == 0:
        global TRAFFIC_BYTES_PER_LAUNCH
        # the PMC passes were taken on the default workload only (40 samples, cfg2, rigid receptor, one batch)
        default_workload = args.samples == 40 and args.cfg == "cfg2" and not args.flex and args.ways == 1
        TRAFFIC_BYTES_PER_LAUNCH = _load_traffic() if default_workload else None
        mfma_busy = _load_mfma_busy() if default_workload else {}
        poses = n_total * args.steps / 20.0
        # the dominant kernel = the instantiation with the larger share of the timed region
        kinds = sorted({k for k in prof.kernel}, key=lambda k: -prof.summary(k)[2])
        roof = None
        if kinds:
            def entry(kname):
                n_, fl_, ms_ = prof.summary(kname)
                ach_ = fl_ / n_ / (ms_ / n_ * 1e-3) / 1e12
                exe_ = prof.executed_flops(kname) / n_ / (ms_ / n_ * 1e-3) / 1e12
                return {"kernel": kname, "bound": "mfma", "unit": "TFLOP/s", "peak": FP32_MFMA_PEAK_TFLOPS,
                        "frac": ach_ / FP32_MFMA_PEAK_TFLOPS, "launches": n_, "avg_launch_ms": ms_ / n_, "achieved": ach_,
                        "algorithmic_gflop_per_launch": fl_ / n_ / 1e9, "executed_tflops": exe_,
                        "executed_frac": exe_ / FP32_MFMA_PEAK_TFLOPS, "share_of_wall": ms_ * 1e-3 / elapsed,
                        "traffic": (TRAFFIC_BYTES_PER_LAUNCH or {}).get(kname), "mfma_busy_pmc": mfma_busy.get(kname)}
            dom = entry(kinds[0])
            roof = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": FP32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": dom["achieved"] / FP32_MFMA_PEAK_TFLOPS, "traffic": dom["traffic"],
                    "launches": dom["launches"], "avg_launch_ms": dom["avg_launch_ms"], "mfma_busy_pmc": dom["mfma_busy_pmc"],
                    "algorithmic_gflop_per_launch": dom["algorithmic_gflop_per_launch"],
                    "executed_tflops": dom["executed_tflops"], "executed_frac": dom["executed_frac"],
                    "note": "achieved = ALGORITHMIC FLOPs of the reference formulation (BASELINE.md section 3: 2FH + 2HW + 2C per edge) / "
                            "kernel time; frac > 1 is possible because the kernel does not execute that formulation: the scalar-input "
                            "tensor-product features are factorised per source node (exact fp32 algebra, DESIGN.md section 4), so only "
                            "executed_tflops of fp32 MFMA work are issued (executed_frac = share of the fp32 MFMA peak)",
                    "conv_share_of_wall": sum(prof.summary(k)[2] for k in kinds) * 1e-3 / elapsed,
                    "other_kernels": [entry(k) for k in kinds[1:]]}

            def hbm_entry(kname):   # HBM-bound kernels of the path: algorithmic bytes / HIP-event time against ~8 TB/s
                n_, by_, ms_ = prof_hbm.hbm_summary(kname)
                if n_ == 0:
                    return None
                gbs = by_ / (ms_ * 1e-3) / 1e9
                return {"kernel": kname, "bound": "hbm", "launches": n_, "avg_launch_ms": ms_ / n_, "achieved": gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_mb_per_launch": by_ / n_ / 1e6,
                        "ms_per_step": ms_ / 2.0, "measured": "2 extra steps after the timed region", "traffic": None,
                        "mfma_busy_pmc": mfma_busy.get(kname)}
            roof["other_kernels"] += [e for e in (hbm_entry("ddp_stage_a_mfma_kernel"), hbm_entry("ddp_segment_reduce_kernel")) if e]
        line = {"metric": "ligand poses/sec (40 samples x 20 steps) on 3dpf", "value": poses / elapsed, "unit": "poses/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"3dpf ({sampler.n_l} lig atoms, 139 residues, {sampler.n_a} pocket atoms), "
                                       f"{args.samples} samples/GPU x 20-step schedule, score model {args.cfg} "
                                       f"(ns={kw['ns']} nv={kw['nv']} L={kw['num_conv_layers']}), "
                                       f"flexible_sidechains={args.flex}", "samples_per_gpu": args.samples,
                           "ways": args.ways,
                           "edges_last_step": getattr(sampler, "last_stats", None) or model.last_stats},
                "roofline": roof}
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only (the other ranks must not wait for it)
            line["cpu_baseline"] = cpu_baseline(args, model, kw, complex_graph)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
