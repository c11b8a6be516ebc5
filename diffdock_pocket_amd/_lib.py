"""ctypes binding of libddp_hip.so (the C ABI declared in include/ddp_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `python -m diffdock_pocket_amd.build`.  There is NO
fallback: if the shared object is missing or a symbol is absent, loading raises - the product never routes
through PyTorch eager or the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DDP_HIP_LIB", os.path.join(HERE, "libddp_hip.so"))  # override: diagnostic builds only

DDP_MAX_TASKS, DDP_MAX_BLOCKS, DDP_MAX_SEGS, DDP_MAX_NS, DDP_EDGE_TILE = 9, 4, 3, 64, 64
F_SCALAR_S0, F_DOT, F_SCALAR_S1, F_VEC_S0, F_CROSS = range(5)
FS = 68  # feature-buffer row stride of ddp_conv.hip

DDP_MAX_GEMM_BATCH = 16
EXPORTS = ["ddp_conv_messages", "ddp_conv_rows", "ddp_stage_a_gh", "ddp_stage_a_gh3", "ddp_segment_reduce", "ddp_edge_featurize", "ddp_edge_featurize_jobs", "ddp_torsion_sh", "ddp_stage_a", "ddp_stage_a_h2",
           "ddp_pose_update", "ddp_sidechain_update", "ddp_sde_update", "ddp_radius_count", "ddp_radius_fill", "ddp_knn", "ddp_group_by_key", "ddp_node_linear", "ddp_scan_jobs", "ddp_mark_jobs", "ddp_rowcopy_jobs", "ddp_select_jobs",
           "ddp_gather_rows", "ddp_clean_pair_maps", "ddp_flex_mark", "ddp_fallback_rowmap", "ddp_step_prologue", "ddp_trrot_head", "ddp_tor_head", "ddp_radius_search_jobs", "ddp_group_by_key_jobs", "ddp_set_occupancy_shaping", "ddp_abi_version", "ddp_last_error", "ddp_source_hash"]


class Seg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_off", C.c_int32), ("count", C.c_int32)]


class Block(C.Structure):
    _fields_ = [("U", C.c_int32), ("n", C.c_int32), ("C", C.c_int32), ("out_off", C.c_int32), ("tile0", C.c_int32),
                ("ntiles", C.c_int32), ("nsub", C.c_int32), ("ups", C.c_int32), ("nseg", C.c_int32),
                ("seg", Seg * DDP_MAX_SEGS), ("g_slot", C.c_int32), ("g_col0", C.c_int32)]


class RoleSeg(C.Structure):
    _fields_ = [("block", C.c_int32), ("tile0", C.c_int32), ("tstride", C.c_int32), ("count", C.c_int32), ("round", C.c_int32)]


DDP_CONV32_WAVES, DDP_MAX_ROLE_SEGS = 4, 2


class ConvShape(C.Structure):
    _fields_ = [("f_in", C.c_int32), ("hid", C.c_int32), ("kp1", C.c_int32), ("hp", C.c_int32), ("hs", C.c_int32),
                ("nct1", C.c_int32), ("d_out", C.c_int32), ("nblocks", C.c_int32), ("fbuf_floats", C.c_int32),
                ("g_cols", C.c_int32 * 2), ("blk", Block * DDP_MAX_BLOCKS), ("nrounds", C.c_int32),
                ("nrole", C.c_int32 * DDP_CONV32_WAVES), ("role", (RoleSeg * DDP_MAX_ROLE_SEGS) * DDP_CONV32_WAVES)]


class ConvTask(C.Structure):
    _fields_ = [("x_src", C.c_void_p), ("ldx_src", C.c_int32), ("n_edges", C.c_int32), ("src", C.c_void_p),
                ("eid", C.c_void_p), ("sh", C.c_void_p), ("seg_ptr", C.c_void_p * DDP_MAX_SEGS),
                ("seg_idx", C.c_void_p * DDP_MAX_SEGS), ("seg_ld", C.c_int32 * DDP_MAX_SEGS),
                ("seg_n", C.c_int32 * DDP_MAX_SEGS), ("w1p", C.c_void_p), ("b1p", C.c_void_p), ("w2p", C.c_void_p),
                ("b2p", C.c_void_p), ("msg", C.c_void_p), ("g", C.c_void_p * 2),
                ("pos", C.c_void_p), ("n_edges_dev", C.c_void_p), ("w1h", C.c_void_p), ("w2h", C.c_void_p), ("h2_range_flag", C.c_void_p),
                ("wsh", C.c_void_p), ("bsp", C.c_void_p), ("gh", C.c_void_p * 2), ("gh_fmt", C.c_int32), ("rows_form", C.c_int32), ("rows_bias_k", C.c_int32), ("rows_seg0", C.c_int32), ("rows_seg1", C.c_int32), ("rows_nts", C.c_int32)]


class ReduceSrc(C.Structure):
    _fields_ = [("msg", C.c_void_p), ("rowptr", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p),
                ("n_edges", C.c_int32), ("rowmap", C.c_void_p), ("n_edges_dev", C.c_void_p)]


DDP_MAX_NODE_JOBS, DDP_MAX_NODE_CAT = 8, 16
DDP_MAX_LIST_JOBS = 12
_P, _I = C.c_void_p, C.c_int32


class ScanJob(C.Structure):
    _fields_ = [("n", _I), ("n_dev", _P), ("flag", _P), ("val", _P), ("rowptr", _P), ("base", _I), ("excl", _P), ("excl2", _P),
                ("list", _P), ("total", _P)]


class MarkJob(C.Structure):
    _fields_ = [("idx", _P), ("n", _I), ("n_dev", _P), ("mask", _P)]


class RowcopyJob(C.Structure):
    _fields_ = [("n_rows", _I), ("keep", _P), ("old_rowptr", _P), ("new_rowptr", _P), ("inp", _P * 3), ("out", _P * 3)]


class SelectJob(C.Structure):
    _fields_ = [("n", _I), ("n_dev", _P), ("mask_a", _P), ("idx_a", _P), ("mask_b", _P), ("idx_b", _P), ("out_idx", _P),
                ("pay", _P * 4), ("pay_add", _I * 4), ("out", _P * 4), ("total", _P), ("block_count", _P), ("block_off", _P)]


class RadiusJob(C.Structure):
    _fields_ = [("x", _P), ("x_ptr", _P), ("y", _P), ("y_batch", _P), ("ny", _I), ("r", C.c_float), ("max_neighbors", _I),
                ("flags", _I), ("graph_div", _P), ("counts", _P), ("offsets", _P), ("base", _I), ("total", _P), ("out_query", _P),
                ("out_x", _P), ("capacity", _I), ("overflow", _P)]


class SdeArgs(C.Structure):
    _fields_ = [("score", _P * 4), ("z", _P * 4), ("out", _P * 4), ("n", _I * 4)]


class GroupJob(C.Structure):
    _fields_ = [("key", _P), ("n_items", _I), ("n_items_dev", _P), ("n_keys", _I), ("pay", _P * 3), ("rowptr", _P), ("perm", _P),
                ("out_key", _P), ("out", _P * 3), ("key_map", _P), ("scratch", _P)]



DDP_MAX_FEATURIZE_JOBS = 8


class FeaturizeJob(C.Structure):
    """ddp_featurize_job_t of include/ddp_hip.h."""
    _fields_ = [("pos_a", _P), ("ia", _P), ("pos_b", _P), ("ib", _P), ("n_edges", _I), ("n_edges_dev", _P), ("offset", _P), ("k_rbf", _I),
                ("coeff", C.c_float), ("pre", _P), ("pre_idx", _P), ("ld_pre", _I), ("pre2", _P), ("n_pre2", _I), ("ld_pre2", _I),
                ("w1d", _P), ("w2", _P), ("b2", _P), ("ns", _I), ("out", _P), ("sh", _P)]


class _BondJob(C.Structure):
    _fields_ = [("pos", _P), ("b0", _P), ("b1", _P), ("n", _I), ("mid", _P), ("vec", _P)]


class _CopyJob(C.Structure):
    _fields_ = [("src", _P), ("dst", _P), ("n", _I)]


class PrologueArgs(C.Structure):
    """ddp_prologue_args_t of include/ddp_hip.h."""
    _fields_ = [("t", _P * 4), ("t_stride", _I * 4), ("sig_min", C.c_float * 4), ("sig_max", C.c_float * 4), ("sigma", _P * 4),
                ("n_graphs", _I), ("cut", _P), ("cut_mul", C.c_float), ("cut_add", C.c_float), ("graph_emb", _P), ("sd", _I),
                ("emb_scale", C.c_float), ("freq", _P), ("lig_pos", _P), ("graph_ptr", _P), ("center", _P),
                ("bonds", _BondJob * 2), ("copy", _CopyJob * 2)]


class TrRotArgs(C.Structure):
    """ddp_trrot_args_t of include/ddp_hip.h."""
    _fields_ = [("gp", _P), ("ld_gp", _I), ("n_graphs", _I), ("ns", _I), ("sd", _I), ("graph_emb", _P), ("w1", _P * 2), ("b1", _P * 2),
                ("w2", _P * 2), ("b2", _P * 2), ("sigma", _P * 2), ("so3_table", _P), ("so3_n", _I), ("so3_lo", C.c_float),
                ("so3_span", C.c_float), ("out", _P * 2)]


class TorArgs(C.Structure):
    """ddp_tor_args_t of include/ddp_hip.h."""
    _fields_ = [("h", _P), ("ld_h", _I), ("n_bonds", _I), ("ns", _I), ("w1", _P), ("w2", _P), ("sigma", _P), ("graph_of_bond", _P),
                ("torus_table", _P), ("torus_n", _I), ("torus_lo", C.c_float), ("torus_span", C.c_float), ("out", _P)]


class NodeJob(C.Structure):
    """ddp_node_job_t of include/ddp_hip.h."""
    _fields_ = [("n_rows", C.c_int32), ("cat", C.c_void_p), ("ld_cat", C.c_int32), ("n_cat", C.c_int32), ("table", C.c_void_p),
                ("feat_off", C.c_int32 * DDP_MAX_NODE_CAT), ("emb_dim", C.c_int32), ("emb_mode", C.c_int32),
                ("dense", C.c_void_p * 2), ("ld_dense", C.c_int32 * 2), ("n_dense", C.c_int32 * 2),
                ("t", C.c_void_p), ("t_stride", C.c_int32), ("scale", C.c_float), ("freq", C.c_void_p),
                ("sig_emb", C.c_void_p), ("ld_sig", C.c_int32), ("sd", C.c_int32), ("sig_out", C.c_void_p), ("ld_sig_out", C.c_int32),
                ("w", C.c_void_p), ("bias", C.c_void_p),
                ("out", C.c_void_p), ("ld_out", C.c_int32), ("ncols", C.c_int32), ("zero_to", C.c_int32),
                ("add", C.c_void_p), ("ld_add", C.c_int32)]


def g_ld(hid: int, gcols: int) -> int:
    """DDP_G_LD of include/ddp_hip.h: floats per node of a G array (rows start on 128-byte boundaries)."""
    return (((hid + 3) // 4 * 4 + 1) * gcols + 31) // 32 * 32


class DdpError(RuntimeError):
    pass


_lib = None


def load():
    """Load libddp_hip.so (once).  Raises if it is missing: there is no non-HIP product path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DdpError(f"{LIB_PATH} not found: build it with `python -m diffdock_pocket_amd.build` "
                       f"(hipcc --offload-arch=gfx950); there is no fallback path")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise DdpError(f"{LIB_PATH} does not export {name}")
    lib.ddp_abi_version.restype = C.c_int
    lib.ddp_last_error.restype = C.c_char_p
    lib.ddp_conv_messages.argtypes = [C.POINTER(ConvShape), C.POINTER(ConvTask), C.c_int, C.c_void_p]
    lib.ddp_conv_messages.restype = C.c_int
    lib.ddp_conv_rows.argtypes = lib.ddp_conv_messages.argtypes
    lib.ddp_conv_rows.restype = C.c_int
    lib.ddp_segment_reduce.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(ReduceSrc), C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]
    lib.ddp_segment_reduce.restype = C.c_int
    lib.ddp_edge_featurize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ddp_edge_featurize.restype = C.c_int
    lib.ddp_edge_featurize_jobs.argtypes = [C.POINTER(FeaturizeJob), C.c_int, C.c_void_p]
    lib.ddp_edge_featurize_jobs.restype = C.c_int
    lib.ddp_torsion_sh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.ddp_torsion_sh.restype = C.c_int
    lib.ddp_stage_a.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.ddp_stage_a.restype = C.c_int
    lib.ddp_stage_a_h2.argtypes = lib.ddp_stage_a.argtypes[:-1] + [C.c_void_p, C.c_void_p]
    lib.ddp_stage_a_h2.restype = C.c_int
    lib.ddp_stage_a_gh.argtypes = lib.ddp_stage_a_h2.argtypes[:-1] + [C.c_void_p, C.c_void_p]
    lib.ddp_stage_a_gh.restype = C.c_int
    lib.ddp_stage_a_gh3.argtypes = lib.ddp_stage_a_gh.argtypes
    lib.ddp_stage_a_gh3.restype = C.c_int
    lib.ddp_set_occupancy_shaping.argtypes = [C.c_int, C.c_int]
    lib.ddp_set_occupancy_shaping.restype = C.c_int
    lib.ddp_pose_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ddp_pose_update.restype = C.c_int
    lib.ddp_sidechain_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
    lib.ddp_sidechain_update.restype = C.c_int
    lib.ddp_radius_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p]
    lib.ddp_radius_count.restype = C.c_int
    lib.ddp_radius_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ddp_radius_fill.restype = C.c_int
    lib.ddp_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ddp_knn.restype = C.c_int
    lib.ddp_group_by_key.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 11
    lib.ddp_group_by_key.restype = C.c_int
    for name, job in (("ddp_scan_jobs", ScanJob), ("ddp_mark_jobs", MarkJob), ("ddp_rowcopy_jobs", RowcopyJob),
                      ("ddp_select_jobs", SelectJob), ("ddp_radius_search_jobs", RadiusJob), ("ddp_group_by_key_jobs", GroupJob)):
        getattr(lib, name).argtypes = [C.POINTER(job), C.c_int, C.c_void_p]
        getattr(lib, name).restype = C.c_int
    lib.ddp_sde_update.argtypes = [C.c_void_p, C.POINTER(SdeArgs), C.c_void_p]
    lib.ddp_sde_update.restype = C.c_int
    lib.ddp_gather_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.ddp_gather_rows.restype = C.c_int
    lib.ddp_clean_pair_maps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    lib.ddp_clean_pair_maps.restype = C.c_int
    lib.ddp_flex_mark.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p]
    lib.ddp_flex_mark.restype = C.c_int
    lib.ddp_fallback_rowmap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ddp_fallback_rowmap.restype = C.c_int
    for name, st in (("ddp_step_prologue", PrologueArgs), ("ddp_trrot_head", TrRotArgs), ("ddp_tor_head", TorArgs)):
        getattr(lib, name).argtypes = [C.POINTER(st), C.c_void_p]
        getattr(lib, name).restype = C.c_int
    lib.ddp_node_linear.argtypes = [C.POINTER(NodeJob), C.c_int, C.c_void_p]
    lib.ddp_node_linear.restype = C.c_int
    if lib.ddp_abi_version() != 17:
        raise DdpError("libddp_hip.so ABI version mismatch")
    lib.ddp_source_hash.restype = C.c_char_p
    if "DDP_HIP_LIB" not in os.environ:   # (diagnostic builds loaded through DDP_HIP_LIB carry extra -D flags, same sources)
        from .build import source_hash
        have, want = lib.ddp_source_hash().decode(), source_hash()
        if have != want:
            raise DdpError(f"{LIB_PATH} was built from other sources (library {have}, tree {want}): rebuild it with "
                           f"`python -m diffdock_pocket_amd.build`")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise DdpError(f"{what} failed (rc={rc}): {load().ddp_last_error().decode()}")
