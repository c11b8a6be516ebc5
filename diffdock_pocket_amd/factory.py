"""`get_model` counterpart of reference utils/utils.py:59-113 for the all-atom score / confidence model: maps the
hyper-parameter namespace stored in `model_parameters.yml` (inference.py:332-336) to constructor kwargs with the same
`in`-guards and defaults.  tests/test_host_logic.py::test_get_model_passes_the_reference_kwargs pins every kwarg to what
the reference's own get_model passes for the README score / confidence settings (oracle/make_golden_factory.py)."""
from __future__ import annotations

from .diffusion import get_timestep_embedding
from .score_model import TensorProductScoreModel


def _has(args, k):
    """`k in args` of the reference (argparse.Namespace implements __contains__); plain objects fall back to hasattr."""
    try:
        return k in args
    except TypeError:
        return hasattr(args, k)


def model_kwargs(args, device, t_to_sigma, confidence_mode=False):
    """The kwargs reference utils/utils.py:73-108 passes to the all-atom TensorProductScoreModel."""
    emb = get_timestep_embedding(embedding_type=args.embedding_type if _has(args, "embedding_type") else "sinusoidal",
                                 dim=args.sigma_embed_dim,
                                 scale=args.embedding_scale if _has(args, "embedding_scale") else 10000)
    cut = args.rmsd_classification_cutoff if _has(args, "rmsd_classification_cutoff") else None
    return dict(
        t_to_sigma=t_to_sigma, device=device, no_torsion=args.no_torsion, timestep_emb_func=emb,
        num_conv_layers=args.num_conv_layers, lig_max_radius=args.max_radius, scale_by_sigma=args.scale_by_sigma,
        sh_lmax=args.sh_lmax, sigma_embed_dim=args.sigma_embed_dim,
        norm_by_sigma=_has(args, "norm_by_sigma") and args.norm_by_sigma,
        ns=args.ns, nv=args.nv, distance_embed_dim=args.distance_embed_dim,
        cross_distance_embed_dim=args.cross_distance_embed_dim, batch_norm=not args.no_batch_norm,
        dropout=args.dropout, use_second_order_repr=args.use_second_order_repr,
        cross_max_distance=args.cross_max_distance, dynamic_max_cross=args.dynamic_max_cross,
        separate_noise_schedule=args.separate_noise_schedule,
        smooth_edges=args.smooth_edges if _has(args, "smooth_edges") else False,
        odd_parity=args.odd_parity if _has(args, "odd_parity") else False,
        lm_embedding_type="esm",                                   # hard-wired, utils/utils.py:71
        confidence_mode=confidence_mode,
        asyncronous_noise_schedule=args.asyncronous_noise_schedule if _has(args, "asyncronous_noise_schedule") else False,
        affinity_prediction=args.affinity_prediction if _has(args, "affinity_prediction") else False,
        parallel=args.parallel if _has(args, "parallel") else 1,
        num_confidence_outputs=len(cut) + 1 if isinstance(cut, list) else 1,          # utils/utils.py:99-101
        parallel_aggregators=args.parallel_aggregators if _has(args, "parallel_aggregators") else "",
        fixed_center_conv=(not args.not_fixed_center_conv) if _has(args, "not_fixed_center_conv") else False,
        no_aminoacid_identities=args.no_aminoacid_identities if _has(args, "no_aminoacid_identities") else False,
        atom_max_neighbors=args.atom_max_neighbors, flexible_sidechains=args.flexible_sidechains,
        include_miscellaneous_atoms=args.include_miscellaneous_atoms if hasattr(args, "include_miscellaneous_atoms") else False,
        use_old_atom_encoder=args.use_old_atom_encoder if hasattr(args, "use_old_atom_encoder") else True)


def get_model(args, device, t_to_sigma, no_parallel=False, confidence_mode=False):
    if not (_has(args, "all_atoms") and args.all_atoms):
        raise NotImplementedError("only the all-atom score model is provided (the README models are all-atom; the "
                                  "coarse-grained class of models/score_model.py:127-658 is out of scope)")
    if getattr(device, "type", str(device)) == "cuda" and not no_parallel:
        # the reference wraps the model in torch_geometric's single-process DataParallel here (utils/utils.py:110-111);
        # inference always passes no_parallel=True (inference.py:433,446).  Multi-GPU here = one process per GPU with the
        # samples sharded (sampler.Sampler(sample_slice=...), bench.py --gpus N).
        raise NotImplementedError("single-process DataParallel is not provided: pass no_parallel=True and shard the samples "
                                  "over one process per GPU")
    model = TensorProductScoreModel(**model_kwargs(args, device, t_to_sigma, confidence_mode))
    model.to(device)
    return model
