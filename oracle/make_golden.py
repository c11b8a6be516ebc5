"""Generate golden vectors by running the REFERENCE's own model source files (TEST INFRASTRUCTURE, oracle/).

Run in the build container only (needs /root/reference):   python -m oracle.make_golden [case ...]

For every case in oracle/cases.py:
  1. import /root/reference/models/all_atom_score_model.py unchanged under oracle/shim.py,
  2. construct its TensorProductScoreModel with the case's kwargs, load the seeded synthetic weights
     (oracle/weights.py) and put it in eval mode,
  3. run its forward on the case's seeded batch,
  4. store under tests/golden/<case>.pt:  the four outputs, every conv layer's output (forward hooks on
     conv_layers / final_conv / tor_bond_conv / sc_tor_bond_conv: mean|.| and a strided sample), edge counts,
     the state_dict key -> shape table (pins SURVEY Appendix A.7) and checksums of the regenerated inputs.
Only tensors / numbers are stored - no reference source, no pickled code.
"""
import os
import sys

import torch

from . import shim
from .cases import CASES, input_checksums
from .weights import synth_state_dict

OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def sample_of(t, n=64):
    flat = t.reshape(-1)
    if flat.numel() <= n:
        return flat.clone()
    idx = torch.linspace(0, flat.numel() - 1, n).long()
    return flat[idx].clone()


def run_case(case):
    ref = shim.import_reference()
    kw = dict(case.model_kwargs())
    kw.update(case.ctor_extras())
    torch.manual_seed(0)
    model = ref.aa.TensorProductScoreModel(**kw)
    template = model.state_dict()
    sd = synth_state_dict(template, case.weight_seed)
    model.load_state_dict(sd, strict=True)
    model.eval()
    batch = case.make_batch()
    conv_out = {}

    def hook(name):
        def f(mod, inp, out):
            conv_out[name] = out.detach().clone()
        return f

    for i, m in enumerate(model.conv_layers):
        m.register_forward_hook(hook(f"conv_layers.{i}"))
    for name in ("final_conv", "tor_bond_conv", "sc_tor_bond_conv"):
        if hasattr(model, name):
            getattr(model, name).register_forward_hook(hook(name))
    with torch.no_grad():
        res = model(batch)
    if case.confidence_mode:
        outputs = {"confidence": res}
    else:
        tr, rot, tor, sc = res
        outputs = {"tr": tr, "rot": rot, "tor": tor, "sc_tor": sc}
    gold = {
        "case": case.name,
        "outputs": outputs,
        "conv_stats": {k: {"shape": list(v.shape), "mean_abs": float(v.abs().mean()) if v.numel() else 0.0,
                           "sample": sample_of(v)} for k, v in conv_out.items()},
        "state_dict_shapes": {k: list(v.shape) for k, v in template.items()},
        "offsets": {k: v.clone() for k, v in template.items() if k.endswith(".offset")},
        "inputs": input_checksums(batch),
        "edge_counts": {"aa": int(batch["atom", "atom"].edge_index.shape[1])},
    }
    return gold


def main():
    names = sys.argv[1:] or list(CASES)
    os.makedirs(OUT_DIR, exist_ok=True)
    for n in names:
        gold = run_case(CASES[n])
        path = os.path.join(OUT_DIR, f"{n}.pt")
        torch.save(gold, path)
        o = gold["outputs"]
        print(f"{n}: " + " ".join(f"{k} {v.flatten()[:3].tolist()}" for k, v in o.items())
              + f" -> {path} ({os.path.getsize(path) / 1e3:.0f} kB)")


if __name__ == "__main__":
    main()
