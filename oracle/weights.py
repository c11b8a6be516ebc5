"""Deterministic, construction-order-independent synthetic weights (TEST INFRASTRUCTURE, oracle/).

Real checkpoints need network access (reference inference.py:320-330), so parity runs on seeded random
weights.  Each tensor is drawn from its own generator seeded by (seed, crc32(key)), so the reference model
(under oracle/shim.py), the oracle restatement and the HIP model all get bit-identical parameters from the
key/shape list alone - independent of how each of them builds its modules.  BatchNorm running statistics are
randomised on purpose (identity BN would hide bugs).  `*.offset` buffers (GaussianSmearing
centres) are kept as they are.
"""
import math
import zlib

import torch


def synth_tensor(key, shape, seed):
    g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    shape = tuple(shape)

    def uni(lo, hi):
        return torch.rand(shape, generator=g) * (hi - lo) + lo

    if key.endswith("running_var"):
        return uni(0.5, 2.0)
    if key.endswith("running_mean"):
        return torch.randn(shape, generator=g) * 0.1
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if "batch_norm" in key or (len(shape) == 1 and "confidence_predictor" in key and key.endswith("weight")):
        if key.endswith("weight"):
            return uni(0.5, 1.5)
        return torch.randn(shape, generator=g) * 0.1
    if "atom_embedding_list" in key:  # xavier-uniform, reference models/score_model.py:69
        b = math.sqrt(6.0 / (shape[0] + shape[1]))
        return uni(-b, b)
    if key.endswith("weight") and len(shape) == 2:  # nn.Linear default: U(-1/sqrt(in), 1/sqrt(in))
        b = 1.0 / math.sqrt(shape[1])
        return uni(-b, b)
    if key.endswith("bias"):
        return uni(-0.1, 0.1)
    raise KeyError(f"no synthetic rule for {key} {shape}")


def re_bn1d(key):
    return False


def synth_state_dict(template, seed):
    """template: {key: tensor or shape}.  Returns {key: tensor}; '.offset' buffers are copied if tensors are given."""
    out = {}
    for k, v in template.items():
        if k.endswith(".offset"):
            if not torch.is_tensor(v):
                raise ValueError("offset buffers must be provided as tensors")
            out[k] = v.clone()
            continue
        shape = tuple(v.shape) if torch.is_tensor(v) else tuple(v)
        out[k] = synth_tensor(k, shape, seed)
    return out
