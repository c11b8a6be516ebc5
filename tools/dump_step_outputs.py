"""Scores and poses of the first denoising steps of bench.py's jobs, saved for a bitwise comparison between code states
(python tools/dump_step_outputs.py OUT.pt): cfg2 40 samples rigid / flexible, cfg2 5 samples rigid, cfg1 4 samples flexible;
steps at schedule positions 0, 1, 2 and 10.  tests/golden/step_outputs_r02_host_path.pt was written by the round-2 host-driven
forward (exact-size lists, host synchronisations); tests/test_gpu_parity.py::test_device_driven_step_equals_the_host_driven_one
compares the device-driven step with it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

JOBS = [("cfg2", False, 40), ("cfg2", True, 40), ("cfg2", False, 5), ("cfg1", True, 4)]
STEPS = [0, 1, 2, 10]


def run_job(cfg, flex, n, device, sampler_kwargs=None):
    import bench
    from diffdock_pocket_amd.batch import set_time
    from diffdock_pocket_amd.diffusion import get_t_schedule
    from diffdock_pocket_amd.sampler import Sampler, SamplerConfig
    from diffdock_pocket_amd.synthetic import make_3dpf_complex
    model, kw = bench.build_model(cfg, flex, device)
    g = make_3dpf_complex(seed=0, flexible_sidechains=flex)
    smp = Sampler(model, g, n, device, SamplerConfig(inference_steps=20, flexible_sidechains=flex), seed=0, **(sampler_kwargs or {}))
    smp.randomize()
    sched = get_t_schedule(20)
    out = {}
    for t_idx in STEPS:
        # scores of the step's forward on the poses the sampler holds, then the step itself
        t = float(sched[t_idx])
        scores = [o.float().cpu().clone() for o in smp.scores(t)]
        smp.step(t_idx, sched)
        torch.cuda.synchronize()
        out[t_idx] = {"scores": scores, "lig_pos": smp.lig_pos.cpu().clone(), "atom_pos": smp.atom_pos.cpu().clone(),
                      "stats": {k: int(v) for k, v in model.last_stats.items() if k.startswith(("E_", "N_", "B"))}}
    if hasattr(smp, "close"):
        smp.close()     # (a captured 40-sample step holds tens of GB of device memory)
    return out


def main():
    dev = torch.device("cuda:0")
    res = {}
    for cfg, flex, n in JOBS:
        res[f"{cfg}_flex{int(flex)}_n{n}"] = run_job(cfg, flex, n, dev)
        print("done", cfg, flex, n, flush=True)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "step_outputs.pt")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(res, path)
    print("saved", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
