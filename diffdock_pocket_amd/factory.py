"""`get_model` counterpart of reference utils/utils.py:59-113 for the all-atom score model: maps the
hyper-parameter namespace stored in `model_parameters.yml` to constructor kwargs with the same `in`-guards."""
from __future__ import annotations

from .diffusion import get_timestep_embedding
from .score_model import TensorProductScoreModel


def _has(args, k):
    return (k in args) if hasattr(args, "__contains__") else hasattr(args, k)


def _get(args, k, default):
    return getattr(args, k) if _has(args, k) or hasattr(args, k) else default


def get_model(args, device, t_to_sigma, no_parallel=False, confidence_mode=False):
    if not _get(args, "all_atoms", False):
        raise NotImplementedError("only the all-atom score model is provided (README models are all-atom)")
    emb = get_timestep_embedding(embedding_type=_get(args, "embedding_type", "sinusoidal"), dim=args.sigma_embed_dim,
                                 scale=_get(args, "embedding_scale", 10000))
    model = TensorProductScoreModel(
        t_to_sigma=t_to_sigma, device=device, no_torsion=args.no_torsion, timestep_emb_func=emb,
        num_conv_layers=args.num_conv_layers, lig_max_radius=args.max_radius, scale_by_sigma=args.scale_by_sigma,
        sh_lmax=args.sh_lmax, sigma_embed_dim=args.sigma_embed_dim, norm_by_sigma=_get(args, "norm_by_sigma", False),
        ns=args.ns, nv=args.nv, distance_embed_dim=args.distance_embed_dim,
        cross_distance_embed_dim=args.cross_distance_embed_dim, batch_norm=not args.no_batch_norm,
        dropout=args.dropout, use_second_order_repr=args.use_second_order_repr,
        cross_max_distance=args.cross_max_distance, dynamic_max_cross=args.dynamic_max_cross,
        separate_noise_schedule=args.separate_noise_schedule, smooth_edges=_get(args, "smooth_edges", False),
        odd_parity=_get(args, "odd_parity", False), lm_embedding_type="esm", confidence_mode=confidence_mode,
        asyncronous_noise_schedule=_get(args, "asyncronous_noise_schedule", False),
        affinity_prediction=_get(args, "affinity_prediction", False), parallel=_get(args, "parallel", 1),
        parallel_aggregators=_get(args, "parallel_aggregators", ""),
        fixed_center_conv=(not args.not_fixed_center_conv) if _has(args, "not_fixed_center_conv") else False,
        no_aminoacid_identities=_get(args, "no_aminoacid_identities", False),
        atom_max_neighbors=args.atom_max_neighbors, flexible_sidechains=args.flexible_sidechains,
        include_miscellaneous_atoms=_get(args, "include_miscellaneous_atoms", False),
        use_old_atom_encoder=_get(args, "use_old_atom_encoder", True))
    model.to(device)
    return model
