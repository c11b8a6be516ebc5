"""Import the reference's own score-model source files (read-only, by path) under a dependency shim.

TEST INFRASTRUCTURE (oracle/), fixture-generation time only: this module needs /root/reference, which
does not exist on the GPU box.  Nothing under tests/ -m gpu, smoke() or bench.py calls it.

The reference imports e3nn / torch_scatter / torch_cluster / torch_geometric / rdkit / Bio, none of which
is installed.  `import_reference()` pre-seeds sys.modules with
  * real (restated) implementations from oracle/thirdparty.py for the ops the forward actually executes,
  * inert MagicMock modules for packages only touched at import time (rdkit, Bio, torch_geometric, ...),
  * table-backed `utils.so3` / `utils.torus` (the real ones take ~8 min to import and draw from the global
    RNG; their tables were captured once by oracle/make_score_norm_tables.py, together with probe values
    evaluated by the reference's own lookup functions, see tests/test_oracle_golden.py),
  * a `datasets` namespace package that points at the reference (the HuggingFace `datasets` wheel
    installed in this image would otherwise shadow it),
and then imports models.all_atom_score_model etc. from /root/reference unchanged.
"""
import importlib
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

from . import thirdparty as tp
from . import score_norm

REF = os.environ.get("DDP_REFERENCE", "/root/reference")

_MOCKED = [
    "rdkit", "rdkit.Chem", "rdkit.Chem.rdchem", "rdkit.Chem.AllChem", "rdkit.Chem.rdMolTransforms",
    "rdkit.Geometry", "rdkit.RDLogger", "rdkit.Chem.rdMolAlign", "rdkit.Chem.rdmolops",
    "Bio", "Bio.PDB", "Bio.PDB.PDBExceptions", "Bio.PDB.Polypeptide", "Bio.PDB.Selection",
    "torch_geometric", "torch_geometric.data", "torch_geometric.loader", "torch_geometric.utils",
    "torch_geometric.nn", "torch_geometric.nn.data_parallel", "torch_geometric.transforms",
    "torch_geometric.loader.dataloader", "torch_geometric.data.dataset",
    "spyrmsd", "spyrmsd.rmsd", "spyrmsd.molecule", "esm", "wandb", "openmm", "prody",
]


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def import_reference():
    """Returns a namespace with the reference modules: .aa (all_atom_score_model), .score_model, .layers,
    .diffusion_utils, .geometry."""
    if not os.path.isdir(REF):
        raise RuntimeError(f"{REF} not present: the shim only works in the build container")
    if getattr(import_reference, "_cache", None) is not None:
        return import_reference._cache
    for name in _MOCKED:
        sys.modules.setdefault(name, mock.MagicMock(name=name))

    o3 = _module("e3nn.o3", Irreps=tp.Irreps, Irrep=tp.Irrep, spherical_harmonics=tp.spherical_harmonics,
                 FullTensorProduct=tp.FullTensorProduct, FullyConnectedTensorProduct=tp.FullyConnectedTensorProduct)
    nn = _module("e3nn.nn", BatchNorm=tp.BatchNorm)
    e3nn = _module("e3nn", o3=o3, nn=nn)
    e3nn.__path__ = []
    sys.modules.update({"e3nn": e3nn, "e3nn.o3": o3, "e3nn.nn": nn})
    sys.modules["torch_scatter"] = _module("torch_scatter", scatter=tp.scatter, scatter_mean=tp.scatter_mean)
    sys.modules["torch_cluster"] = _module("torch_cluster", radius=tp.radius, radius_graph=tp.radius_graph,
                                           knn_graph=tp.knn_graph)

    # the reference's namespace packages
    for pkg in ("datasets", "utils", "models"):
        old = sys.modules.pop(pkg, None)
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, pkg)]
        sys.modules[pkg] = m
        for k in [k for k in sys.modules if k.startswith(pkg + ".")]:
            del sys.modules[k]
        del old

    tables = score_norm.ScoreNormTables.load()
    so3 = _module("utils.so3", score_norm=lambda eps: torch.from_numpy(
        tables.so3_score_norm(eps.numpy() if torch.is_tensor(eps) else np.asarray(eps))).float())
    torus = _module("utils.torus", score_norm=lambda sigma: tables.torus_score_norm(np.asarray(sigma)))
    sys.modules["utils.so3"] = so3
    sys.modules["utils.torus"] = torus
    sys.modules["utils"].so3 = so3
    sys.modules["utils"].torus = torus

    ns = types.SimpleNamespace()
    ns.aa = importlib.import_module("models.all_atom_score_model")
    ns.score_model = importlib.import_module("models.score_model")
    ns.layers = importlib.import_module("models.layers")
    ns.diffusion_utils = importlib.import_module("utils.diffusion_utils")
    ns.geometry = importlib.import_module("utils.geometry")
    ns.torsion = importlib.import_module("utils.torsion")
    import_reference._cache = ns
    return ns
