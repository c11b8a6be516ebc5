"""Shared test helpers: load a golden case, rebuild its seeded weights and inputs."""
import os

import torch

from oracle.cases import CASES, input_checksums
from oracle.weights import synth_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, f"{name}.pt"), weights_only=False)


def golden_state_dict(gold, seed):
    template = {k: tuple(v) for k, v in gold["state_dict_shapes"].items()}
    template.update(gold["offsets"])
    return synth_state_dict(template, seed)


def case_inputs(name):
    case = CASES[name]
    gold = load_golden(name)
    batch = case.make_batch()
    chk = input_checksums(batch)
    for k, v in gold["inputs"].items():
        assert abs(chk[k] - v) <= 1e-6 * max(1.0, abs(v)), f"regenerated input {k} differs from the golden run"
    return case, gold, batch, golden_state_dict(gold, case.weight_seed)


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): relative to the largest reference component (fp32 parity metric)."""
    if a.numel() == 0 and b.numel() == 0:
        return 0.0
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
