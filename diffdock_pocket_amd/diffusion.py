"""Noise schedules and time embeddings used inside the score-model forward.

Counterparts of reference utils/diffusion_utils.py:22-34 (`t_to_sigma`), :73-84 (`sinusoidal_embedding`),
:104-109 (`get_timestep_embedding`), :112-117 (`get_t_schedule`).  Plain PyTorch host code (north_star keeps
this side in Python); the values feed the HIP kernels as device tensors.
"""
from __future__ import annotations

import functools
import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class SigmaRanges:
    """sigma_min / sigma_max per component; defaults = reference utils/parsing.py:86-93 + README.md:72."""
    tr_sigma_min: float = 0.1
    tr_sigma_max: float = 5.0
    rot_sigma_min: float = 0.03
    rot_sigma_max: float = 1.55
    tor_sigma_min: float = 0.03
    tor_sigma_max: float = 3.14
    sidechain_tor_sigma_min: float = 0.03
    sidechain_tor_sigma_max: float = 3.14


def t_to_sigma(t_tr, t_rot, t_tor, t_sc_tor, args: SigmaRanges):
    """sigma = sigma_min^(1-t) * sigma_max^t for each of the four components."""
    def one(t, lo, hi):
        return lo ** (1 - t) * hi ** t
    return (one(t_tr, args.tr_sigma_min, args.tr_sigma_max),
            one(t_rot, args.rot_sigma_min, args.rot_sigma_max),
            one(t_tor, args.tor_sigma_min, args.tor_sigma_max),
            one(t_sc_tor, args.sidechain_tor_sigma_min, args.sidechain_tor_sigma_max))


@functools.lru_cache(maxsize=16)
def _frequencies(half: int, max_positions: int, device):
    """w_k of sinusoidal_embedding (a constant of the embedding size: built once per device, not per call).  Evaluated on the
    CPU and uploaded: the arguments scale * t * w_k reach ~1000, where one ulp of w_k (a device exp against the host's) moves
    sin / cos by 3e-5 - the parity target is the reference's CPU path."""
    return torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(max_positions) / (half - 1))).to(device)


def sinusoidal_embedding(timesteps: torch.Tensor, dim: int, scale: float = 1.0, max_positions: int = 10000):
    """[sin(s*t*w_k), cos(s*t*w_k)], w_k = exp(-k ln(max_positions)/(half-1)); zero-padded if dim is odd."""
    assert timesteps.dim() == 1
    half = dim // 2
    freq = _frequencies(half, max_positions, timesteps.device)
    arg = scale * timesteps.float()[:, None] * freq[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb


def get_timestep_embedding(embedding_type: str, dim: int, scale: float = 10000):
    if embedding_type != "sinusoidal":
        raise NotImplementedError("only the sinusoidal timestep embedding is supported (README.md:72 config)")
    return functools.partial(sinusoidal_embedding, dim=dim, scale=scale)


def get_t_schedule(inference_steps: int, t_max: float = 1.0):
    """'expbeta' schedule with alpha=beta=1 (reference default) == linspace(t_max, 0, steps+1)[:-1]."""
    return np.linspace(t_max, 0, inference_steps + 1)[:-1]
