"""diffdock_pocket_amd - MI355X-native (gfx950) drop-in for DiffDock-Pocket's reverse-diffusion score-model
hot path.  See DESIGN.md for the scope, include/ddp_hip.h for the C ABI and INTEGRATION.md for how the reference
binds to it."""
from .batch import HeteroBatch, Store, collate, set_time  # noqa: F401
from .diffusion import SigmaRanges, get_t_schedule, get_timestep_embedding, sinusoidal_embedding, t_to_sigma  # noqa: F401

__all__ = ["HeteroBatch", "Store", "collate", "set_time", "SigmaRanges", "get_t_schedule", "get_timestep_embedding",
           "sinusoidal_embedding", "t_to_sigma", "get_model", "TensorProductScoreModel"]


def __getattr__(name):  # lazy: importing the model pulls in torch.nn and the ctypes binding
    if name == "TensorProductScoreModel":
        from .score_model import TensorProductScoreModel
        return TensorProductScoreModel
    if name == "get_model":
        from .factory import get_model
        return get_model
    raise AttributeError(name)
