"""Golden kwargs of the reference's own `get_model` (utils/utils.py:59-113) for the README settings (TEST INFRASTRUCTURE;
build container only: needs /root/reference).   python -m oracle.make_golden_factory

The hyper-parameter namespaces are produced by the reference's OWN argument parsers fed with the README command lines
(README.md:72 big score model via utils/parsing.py::parse_train_args; README.md:88 confidence model via the module-level parser
of filtering/filtering_train.py), get_model is the reference's unmodified function with the two model classes replaced by a
recorder, and what it passes is stored as plain JSON (tests/golden/factory_kwargs.json): values only, the time-embedding
function as its values on a probe."""
import argparse
import importlib
import json
import os
import shlex
import sys
import types

import torch

from . import shim

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "factory_kwargs.json")

# README.md:72 (flags after `python -m train`) and README.md:88 (after `python -m filtering.filtering_train`)
SCORE_CMD = ("--run_name big_score_model --test_sigma_intervals --log_dir workdir --lr 1e-3 --tr_sigma_min 0.1 --tr_sigma_max 5 "
             "--rot_sigma_min 0.03 --rot_sigma_max 1.55 --tor_sigma_min 0.03 --sidechain_tor_sigma_min 0.03 --batch_size 16 --ns 60 "
             "--nv 10 --num_conv_layers 6 --distance_embed_dim 64 --cross_distance_embed_dim 64 --sigma_embed_dim 64 "
             "--dynamic_max_cross --scheduler plateau --scale_by_sigma --dropout 0.1 --sampling_alpha 1 --sampling_beta 1 --remove_hs "
             "--c_alpha_max_neighbors 24 --atom_max_neighbors 8 --receptor_radius 15 --num_dataloader_workers 1 --cudnn_benchmark "
             "--rot_alpha 1 --rot_beta 1 --tor_alpha 1 --tor_beta 1 --val_inference_freq 5 --use_ema --scheduler_patience 30 "
             "--n_epochs 750 --all_atom --sh_lmax 1 --split_train data/splits/timesplit_no_lig_overlap_train "
             "--split_val data/splits/timesplit_no_lig_overlap_val_aligned --pocket_reduction --pocket_buffer 10 --flexible_sidechains "
             "--flexdist 3.5 --flexdist_distance_metric prism --protein_file protein_esmfold_aligned_tr_fix --compare_true_protein "
             "--conformer_match_sidechains --conformer_match_score exp --match_max_rmsd 2 --use_original_conformer_fallback "
             "--use_original_conformer")
CONF_CMD = ("--run_name confidence_model --original_model_dir workdir/small_score_model --ckpt best_ema_inference_epoch_model.pt "
            "--inference_steps 20 --samples_per_complex 7 --batch_size 16 --n_epochs 100 --lr 3e-4 --scheduler_patience 50 --ns 24 "
            "--nv 6 --num_conv_layers 5 --dynamic_max_cross --scale_by_sigma --dropout 0.1 --all_atoms --sh_lmax 1 "
            "--split_train data/splits/timesplit_no_lig_overlap_train --split_val data/splits/timesplit_no_lig_overlap_val_aligned "
            "--log_dir workdir --cache_path .cache/data_filtering --data_dir data/PDBBIND_atomCorrected --remove_hs "
            "--c_alpha_max_neighbors 24 --receptor_radius 15 --esm_embeddings_path data/esm2_3billion_embeddings.pt "
            "--main_metric loss --main_metric_goal min --best_model_save_frequency 5 --rmsd_classification_cutoff 2 "
            "--sc_rmsd_classification_cutoff 1 --protein_file protein_esmfold_aligned_tr_fix --use_original_model_cache "
            "--pocket_reduction --pocket_buffer 10 --cache_creation_id 1 --cache_ids_to_combine 1 2 3 4")
# a third namespace: two classification cutoffs -> num_confidence_outputs = 3 (utils/utils.py:99-101), old yml without the
# newer keys (exercises the `in` guards: not_fixed_center_conv / use_old_atom_encoder absent)
CONF2_CMD = CONF_CMD.replace("--rmsd_classification_cutoff 2", "--rmsd_classification_cutoff 2 5")

PROBE_T = [0.0, 0.1, 0.5, 1.0]


class _MockFinder:
    """Any not-yet-seeded submodule of the packages the shim mocks (Bio.PDB.Model, rdkit.Chem.Descriptors, ...) is an inert
    MagicMock too: filtering/filtering_train.py pulls in the whole data pipeline at import time."""
    PREFIXES = ("rdkit", "Bio", "torch_geometric", "spyrmsd", "esm", "wandb", "openmm", "prody", "pdbfixer", "biopandas", "posebusters")

    def find_spec(self, name, path=None, target=None):
        import importlib.machinery
        from unittest import mock
        if name.split(".")[0] in self.PREFIXES:
            mod = mock.MagicMock(name=name)
            mod.__path__ = []
            spec = importlib.machinery.ModuleSpec(name, _MockLoader(mod), is_package=True)
            mod.__spec__ = spec
            return spec
        return None


class _MockLoader:
    def __init__(self, mod):
        self.mod = mod

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        pass


class Recorder:
    last = None

    def __init__(self, **kw):
        Recorder.last = kw

    def to(self, device):
        return self


def jsonable(v):
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    return repr(v)


def record(utils_mod, args, confidence_mode):
    t2s = "T_TO_SIGMA_SENTINEL"
    utils_mod.get_model(args, torch.device("cpu"), t_to_sigma=t2s, no_parallel=True, confidence_mode=confidence_mode)
    kw = dict(Recorder.last)
    assert kw.pop("t_to_sigma") == t2s
    assert kw.pop("device") == torch.device("cpu")
    emb = kw.pop("timestep_emb_func")
    return {"args": {k: jsonable(v) for k, v in vars(args).items()}, "kwargs": {k: jsonable(v) for k, v in kw.items()},
            "timestep_emb_probe": {"t": PROBE_T, "values": emb(torch.tensor(PROBE_T)).tolist()}}


def main():
    shim.import_reference()
    from unittest import mock
    import importlib.machinery
    for k in list(sys.modules):   # the shim's seeded mocks become packages so that their submodules resolve through the finder
        if k.split(".")[0] in _MockFinder.PREFIXES and isinstance(sys.modules[k], mock.MagicMock):
            sys.modules[k].__path__ = []
            sys.modules[k].__spec__ = importlib.machinery.ModuleSpec(k, _MockLoader(sys.modules[k]), is_package=True)
    sys.meta_path.insert(0, _MockFinder())
    ref_root = shim.REF
    m = types.ModuleType("filtering")
    m.__path__ = [os.path.join(ref_root, "filtering")]
    sys.modules["filtering"] = m
    for name in ("yaml",):
        importlib.import_module(name)
    sys.modules.setdefault("sklearn.metrics", importlib.import_module("sklearn.metrics"))
    utils_mod = importlib.import_module("utils.utils")
    utils_mod.AAScoreModel = Recorder
    utils_mod.CGScoreModel = None
    out = {}
    parsing = importlib.import_module("utils.parsing")
    argv0 = sys.argv
    try:
        sys.argv = ["train"] + shlex.split(SCORE_CMD)
        out["score_README_72"] = record(utils_mod, parsing.parse_train_args(), False)
        for key, cmd in (("confidence_README_88", CONF_CMD), ("confidence_two_cutoffs", CONF2_CMD)):
            sys.argv = ["filtering_train"] + shlex.split(cmd)
            for k in [k for k in sys.modules if k.startswith("filtering.")]:
                del sys.modules[k]
            ft = importlib.import_module("filtering.filtering_train")
            a = ft.args
            # "Sidechain configuration is specified by the original model" (filtering/filtering_train.py:473-476): the small
            # score model of README.md:82 has --flexible_sidechains --flexdist 3.5 --flexdist_distance_metric prism
            a.flexible_sidechains, a.flexdist, a.flexdist_distance_metric = True, 3.5, "prism"
            out[key] = record(utils_mod, a, True)
    finally:
        sys.argv = argv0
    # an old-style yml namespace without the newer keys (the `in`-guards' other branch)
    old = argparse.Namespace(**{k: v for k, v in out["score_README_72"]["args"].items()
                                if k not in ("not_fixed_center_conv", "use_old_atom_encoder", "include_miscellaneous_atoms",
                                             "embedding_scale", "embedding_type", "smooth_edges", "odd_parity", "norm_by_sigma",
                                             "asyncronous_noise_schedule", "affinity_prediction", "parallel", "parallel_aggregators",
                                             "no_aminoacid_identities")})
    out["score_old_yml"] = record(utils_mod, old, False)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in out.items():
        print(k, {kk: vv for kk, vv in v["kwargs"].items() if kk in ("ns", "nv", "num_conv_layers", "num_confidence_outputs",
                                                                   "fixed_center_conv", "use_old_atom_encoder", "confidence_mode")})
    print("wrote", OUT)


if __name__ == "__main__":
    main()
