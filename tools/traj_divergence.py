"""Diagnostic: divergence (angstrom) between the HIP sampler and the oracle-driven CPU sampler over a whole cfg1 job
(BASELINE configs[0]: 4 samples x 20 steps), per step, for several scalings of the synthetic conv fc weights.
Calibrates the bound of tests/test_gpu_parity.py::test_cfg1_job_end_to_end_against_the_cpu_sampler."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import case_inputs  # noqa: E402
from oracle.cases import CASES  # noqa: E402
from oracle.ref_model import OracleScoreModel  # noqa: E402
from diffdock_pocket_amd.diffusion import get_t_schedule  # noqa: E402
from diffdock_pocket_amd.sampler import Sampler, SamplerConfig  # noqa: E402
from diffdock_pocket_amd.score_model import TensorProductScoreModel  # noqa: E402
from diffdock_pocket_amd.synthetic import make_3dpf_complex  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    case = CASES["cfg1_full"]
    _, _, _, sd0 = case_inputs(case.name)
    g = make_3dpf_complex(seed=0, flexible_sidechains=True)
    sched = get_t_schedule(20)
    cfg = SamplerConfig(inference_steps=20, flexible_sidechains=True)
    for scale in (1.0, 0.5, 0.25):
        sd = {k: (v * scale if (".fc." in k and k.endswith("weight")) else v) for k, v in sd0.items()}
        kw = dict(case.model_kwargs())
        kw.update(case.ctor_extras())
        kw["device"] = dev
        model = TensorProductScoreModel(**kw)
        model.load_state_dict(sd, strict=True)
        model = model.to(dev).eval()
        oracle = OracleScoreModel(case.oracle_config(), sd)
        for seed in (9, 3):
            s_gpu = Sampler(model, g, 4, dev, cfg, seed=seed)
            s_cpu = Sampler(lambda b: oracle(b), g, 4, torch.device("cpu"), cfg, seed=seed)
            s_gpu.randomize()
            s_cpu.randomize()
            start = s_cpu.lig_pos.clone()
            out = []
            for i in range(20):
                s_gpu.step(i, sched)
                s_cpu.step(i, sched)
                dl = float((s_gpu.lig_pos.cpu() - s_cpu.lig_pos).abs().max())
                da = float((s_gpu.atom_pos.cpu() - s_cpu.atom_pos).abs().max())
                out.append((dl, da))
            moved = float((s_cpu.lig_pos - start).abs().max())
            print(f"scale {scale} seed {seed}: moved {moved:.2f} A; max |d lig|, |d atom| per step: " +
                  " ".join(f"{a:.1e}/{b:.1e}" for a, b in out), flush=True)


if __name__ == "__main__":
    main()
