"""Reverse-diffusion sampling loop around the score model: the caller of the hot path.

Counterpart of reference utils/sampling.py:16-60 (`randomize_position`) and :70-286 (`sampling`) for the default
inference settings (SDE with low-temperature sampling, no SVGD, no confidence model), restructured for the
MI355X: the N sample graphs of one complex are collated ONCE into a device-resident batch (the reference
re-collates a python list on the CPU every step, utils/sampling.py:100,112-114) and the pose update
(utils/diffusion_utils.py:37-70, utils/torsion.py:68-94,251-278, utils/geometry.py:72-86,209-243) is batched over
the N samples in PyTorch-ROCm instead of a per-sample numpy/scipy loop.  North-star keeps this side in Python.

Random numbers are drawn for ALL n_total samples of the job from one seeded CPU generator and sliced by
`sample_slice`, so results do not depend on how samples are sharded over GPUs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from .batch import HeteroBatch, collate, set_time
from .diffusion import SigmaRanges, get_t_schedule

# defaults of reference inference.py:93-101
TEMP_SAMPLING = (0.9766350103728372, 6.077432837220868, 6.761568162335063, 1.4487910576602347)
TEMP_PSI = (1.5102572175711826, 0.8141168207563049, 0.7661845361370018, 1.339614553802453)
TEMP_SIGMA_DATA = 0.48884149503636976


def rotvec_to_matrix(v: torch.Tensor) -> torch.Tensor:
    """Rodrigues formula, [...,3] -> [...,3,3]; equals scipy Rotation.from_rotvec(v).as_matrix() and
    reference utils/geometry.py:72-86 (axis_angle_to_matrix)."""
    ang = v.norm(dim=-1, keepdim=True)
    small = ang < 1e-6
    safe = torch.where(small, torch.ones_like(ang), ang)
    a = torch.where(small, 1.0 - ang * ang / 6.0, torch.sin(safe) / safe)                 # sin(t)/t
    b = torch.where(small, 0.5 - ang * ang / 24.0, (1.0 - torch.cos(safe)) / (safe * safe))  # (1-cos t)/t^2
    x, y, z = v.unbind(-1)
    zero = torch.zeros_like(x)
    K = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], -1).reshape(v.shape[:-1] + (3, 3))
    eye = torch.eye(3, dtype=v.dtype, device=v.device).expand(K.shape)
    return eye + a.unsqueeze(-1) * K + b.unsqueeze(-1) * (K @ K)


def kabsch(A: torch.Tensor, B: torch.Tensor):
    """Batched rigid_transform_Kabsch_3D_torch (reference utils/geometry.py:209-243): A, B [N, n, 3] point sets;
    returns R [N,3,3], t [N,1,3] with  A @ R^T + t ~ B."""
    ca, cb = A.mean(1, keepdim=True), B.mean(1, keepdim=True)
    H = (A - ca).transpose(1, 2) @ (B - cb)
    U, S, Vt = torch.linalg.svd(H)
    R = Vt.transpose(1, 2) @ U.transpose(1, 2)
    neg = torch.linalg.det(R) < 0          # reflection case (:232-236); applied without a host-side branch (no sync)
    D = torch.ones(3, dtype=A.dtype, device=A.device)
    D[2] = -1.0
    R = torch.where(neg.reshape(-1, 1, 1), (Vt.transpose(1, 2) * D) @ U.transpose(1, 2), R)
    t = cb - ca @ R.transpose(1, 2)
    return R, t


def rotate_index_lists(mask_rotate: torch.Tensor):
    """mask_rotate [T,n] bool -> per-bond index tensors of the atoms that move.  Built once: indexing with the boolean
    mask itself costs a device-to-host synchronisation (nonzero) per bond and step."""
    return [mask_rotate[j].nonzero(as_tuple=True)[0] for j in range(mask_rotate.shape[0])]


def apply_torsions(pos: torch.Tensor, bonds: torch.Tensor, mask_rotate, angles: torch.Tensor):
    """Batched modify_conformer_torsion_angles (reference utils/torsion.py:68-94).
    pos [N,n,3]; bonds [T,2] (u,v); mask_rotate [T,n] bool or rotate_index_lists(mask_rotate); angles [N,T]."""
    pos = pos.clone()
    for j in range(bonds.shape[0]):
        u, v = int(bonds[j, 0]), int(bonds[j, 1])
        axis = pos[:, u] - pos[:, v]
        rot = rotvec_to_matrix(axis * (angles[:, j:j + 1] / axis.norm(dim=-1, keepdim=True)))
        m = mask_rotate[j]
        pv = pos[:, v:v + 1]
        pos[:, m] = (pos[:, m] - pv) @ rot.transpose(1, 2) + pv
    return pos


def apply_sidechain_torsions(pos, edge_idx, subcomponents, mapping, angles):
    """Batched modify_sidechains (reference utils/diffusion_utils.py:63-70, utils/torsion.py:251-278): bonds are
    applied sequentially in list order.  pos [N,n,3]; angles [N,S]."""
    pos = pos.clone()
    for j in range(edge_idx.shape[0]):
        u, v = int(edge_idx[j, 0]), int(edge_idx[j, 1])
        idx = subcomponents[int(mapping[j, 0]):int(mapping[j, 1])]
        axis = pos[:, u] - pos[:, v]
        rot = rotvec_to_matrix(axis * (angles[:, j:j + 1] / axis.norm(dim=-1, keepdim=True)))
        pv = pos[:, v:v + 1]
        pos[:, idx] = (pos[:, idx] - pv) @ rot.transpose(1, 2) + pv
    return pos


def apply_sidechain_torsions_hip(pos, edge_idx_i32, sub_i32, map_i32, angles):
    """apply_sidechain_torsions for a device-resident batch in one launch (ddp_sidechain_update); returns a NEW tensor (the
    score model's static-graph cache keys on tensor identity / version)."""
    from . import _lib as L
    lib = L.load()
    pos = pos.contiguous()
    out = torch.empty_like(pos)
    angles = angles.contiguous().float()
    L.check(lib.ddp_sidechain_update(pos.data_ptr(), pos.shape[0], pos.shape[1], angles.data_ptr(), angles.shape[1],
                                     edge_idx_i32.data_ptr(), sub_i32.data_ptr(), map_i32.data_ptr(), out.data_ptr(),
                                     torch._C._cuda_getCurrentRawStream(pos.device.index)), "ddp_sidechain_update")
    return out


def modify_conformer_hip(pos, tr, rot, tor, bonds_i32, mask_u8):
    """modify_conformer for a device-resident batch in ONE launch (ddp_pose_update, csrc/ddp_pose.hip); same arguments
    as modify_conformer with bonds as int32 [T,2] and mask_rotate as uint8 [T,n] device tensors."""
    from . import _lib as L
    lib = L.load()
    pos = pos.contiguous()
    out = torch.empty_like(pos)
    tr, rot = tr.contiguous().float(), rot.contiguous().float()
    has_tor = tor is not None and tor.shape[1] > 0
    if has_tor:
        tor = tor.contiguous().float()
    L.check(lib.ddp_pose_update(pos.data_ptr(), pos.shape[0], pos.shape[1], tr.data_ptr(), rot.data_ptr(),
                                tor.data_ptr() if has_tor else None, tor.shape[1] if has_tor else 0,
                                bonds_i32.data_ptr() if has_tor else None, mask_u8.data_ptr() if has_tor else None,
                                out.data_ptr(), torch._C._cuda_getCurrentRawStream(pos.device.index)), "ddp_pose_update")
    return out


def modify_conformer(pos, tr, rot, tor, bonds, mask_rotate):
    """Batched modify_conformer (reference utils/diffusion_utils.py:37-60), pivot=None.
    pos [N,n,3]; tr, rot [N,3]; tor [N,T] or None."""
    center = pos.mean(1, keepdim=True)
    R = rotvec_to_matrix(rot)
    rigid = (pos - center) @ R.transpose(1, 2) + tr.unsqueeze(1) + center
    if tor is None or tor.shape[1] == 0:
        return rigid
    flex = apply_torsions(rigid, bonds, mask_rotate, tor)
    Rk, tk = kabsch(flex, rigid)
    return flex @ Rk.transpose(1, 2) + tk


@dataclass
class SamplerConfig:
    inference_steps: int = 20
    sigma: SigmaRanges = field(default_factory=SigmaRanges)
    temp_sampling: Sequence[float] = TEMP_SAMPLING
    temp_psi: Sequence[float] = TEMP_PSI
    temp_sigma_data: float = TEMP_SIGMA_DATA
    no_final_step_noise: bool = False
    no_random: bool = False
    ode: bool = False
    flexible_sidechains: bool = True
    no_torsion: bool = False
    hip_graph: bool = True     # on a HIP device: capture one denoising step (forward + SDE step + pose update) in a hipGraph
                               # after two ordinary steps and replay it from then on (no host work per launch)


class Sampler:
    """Holds the device-resident batch of `n_local` samples of one complex (samples `sample_slice` of `n_total`)."""

    def __init__(self, model, complex_graph: HeteroBatch, n_total: int, device, cfg: SamplerConfig, seed: int = 0,
                 sample_slice: Optional[slice] = None):
        self.model, self.cfg, self.device = model, cfg, device
        self.n_total = n_total
        self.slice = sample_slice or slice(0, n_total)
        self.n = len(range(*self.slice.indices(n_total)))
        self.gen = torch.Generator().manual_seed(seed)
        g = complex_graph
        self.n_l, self.n_a = g["ligand"].pos.shape[0], g["atom"].pos.shape[0]
        em = g["ligand"].edge_mask.bool()
        self.bonds = g["ligand", "ligand"].edge_index.t()[em].clone()             # [T,2] (u,v)
        mr = g["ligand"].mask_rotate
        self.mask_rotate = torch.as_tensor(np.asarray(mr if isinstance(mr, np.ndarray) else mr[0])).bool().to(device)
        self.rot_idx = rotate_index_lists(self.mask_rotate)
        self.T = int(self.bonds.shape[0])
        self.bonds_i32 = self.bonds.to(torch.int32).contiguous().to(device)
        self.mask_u8 = self.mask_rotate.to(torch.uint8).contiguous()
        self.has_flex = cfg.flexible_sidechains and len(g["flexResidues"]) > 0
        if self.has_flex:
            fr = g["flexResidues"]
            self.sc_edge_idx, self.sc_sub = fr.edge_idx.clone(), fr.subcomponents.clone().to(device)
            self.sc_map = fr.subcomponentsMapping.clone()
            self.S = int(self.sc_edge_idx.shape[0])
            self.sc_edge_i32 = self.sc_edge_idx.to(torch.int32).contiguous().to(device)
            self.sc_sub_i32 = self.sc_sub.to(torch.int32).contiguous()
            self.sc_map_i32 = self.sc_map.to(torch.int32).contiguous().to(device)
        else:
            self.S = 0
        self.batch = collate([g] * self.n).to(device)
        self.lig_pos = self.batch["ligand"].pos.reshape(self.n, self.n_l, 3).clone()
        self.atom_pos = self.batch["atom"].pos.reshape(self.n, self.n_a, 3).clone()
        self.on_hip = torch.device(device).type == "cuda"
        self._graph = None
        self._graph_epoch, self._graph_stats, self._graph_keep = 0, None, None
        self.graph_enabled = True      # False: the step is launched kernel by kernel even if a graph has been captured
        self._steps_run = 0
        if self.on_hip:
            self._init_step_buffers()

    # -- device-resident step state (HIP) ------------------------------------------------------------------------
    def _init_step_buffers(self):
        """One device buffer holds everything that changes from step to step and is not computed on the device: the diffusion
        time, the SDE coefficients (ddp_sde_update) and the step's noise - written by ONE host-to-device copy per step.  The
        batch's time tensors are stride-0 views of the time slot, the poses are updated in place: a step has no argument that
        changes, so it can be captured once and replayed."""
        dev, n = self.device, self.n
        T, S_ = (self.T if not self.cfg.no_torsion else 0), self.S
        self._off = {"t": 0, "coef": 8, "z_tr": 16, "z_rot": 16 + 3 * n, "z_tor": 16 + 6 * n, "z_sc": 16 + 6 * n + n * T}
        self._n_par = 16 + 6 * n + n * T + n * S_
        self.params = torch.zeros(self._n_par, device=dev)
        o = self._off
        self.t_dev = self.params[0:1]
        self.coef = self.params[o["coef"]:o["coef"] + 8]
        self.z_tr = self.params[o["z_tr"]:o["z_tr"] + 3 * n].view(n, 3)
        self.z_rot = self.params[o["z_rot"]:o["z_rot"] + 3 * n].view(n, 3)
        self.z_tor = self.params[o["z_tor"]:o["z_tor"] + n * T].view(n, T) if T > 0 else None
        self.z_sc = self.params[o["z_sc"]:o["z_sc"] + n * S_].view(n, S_) if S_ > 0 else None
        self.upd = {"tr": torch.zeros(n, 3, device=dev), "rot": torch.zeros(n, 3, device=dev),
                    "tor": torch.zeros(n, T, device=dev) if T > 0 else None, "sc": torch.zeros(n, S_, device=dev) if S_ > 0 else None}
        self._views = {}
        self._pin = None
        self._bind_batch()

    def _bind_batch(self):
        """Poses and times of the batch = views of the sampler's device buffers (re-installed if a caller put other tensors
        there, e.g. batch.set_time for a one-off forward)."""
        b = self.batch
        if self._views and all(b[nt].node_t is self._views[nt] for nt in ("ligand", "receptor", "atom")) \
                and b.complex_t is self._views["complex"] and b["ligand"].pos is self._views["lpos"] and b["atom"].pos is self._views["apos"]:
            return
        b["ligand"].pos = self._views["lpos"] = self.lig_pos.view(-1, 3)
        b["atom"].pos = self._views["apos"] = self.atom_pos.view(-1, 3)
        for nt in ("ligand", "receptor", "atom"):
            tn = self.t_dev.expand(b[nt].num_nodes)
            b[nt].node_t = self._views[nt] = {k: tn for k in ("tr", "rot", "tor", "sc_tor")}
        tb = self.t_dev.expand(b.num_graphs)
        b.complex_t = self._views["complex"] = {k: tb for k in ("tr", "rot", "tor", "sc_tor")}
        ts = (b["receptor"].node_t["tr"], b["atom"].node_t["tr"])
        b.ddp_time_hint = (tuple(id(t) for t in ts), tuple(t._version for t in ts), True)    # one time for all nodes (batch.set_time)

    def scores(self, t: float):
        """The score model on the current poses at diffusion time t (no pose update)."""
        if not self.on_hip:
            b = self.batch
            b["ligand"].pos, b["atom"].pos = self.lig_pos.reshape(-1, 3), self.atom_pos.reshape(-1, 3)
            set_time(b, t, t, t, t, device=self.device)
            return self.model(b)
        with torch.cuda.device(self.device):
            self._bind_batch()
            self.t_dev.fill_(t)
            return self.model(self.batch)

    def snapshot(self):
        """(poses, generator state): `restore` puts the sampler back there (bench.py warms up - and captures - on the timed
        sampler itself and then restarts the job from its first step)."""
        return self.lig_pos.clone(), self.atom_pos.clone(), self.gen.get_state()

    def restore(self, snap):
        self._set_pos("lig_pos", snap[0])
        self._set_pos("atom_pos", snap[1])
        self.gen.set_state(snap[2])

    def _step_coefficients(self, t_idx, schedule):
        """(t, [a_tr b_tr a_rot b_rot a_tor b_tor a_sc b_sc], noise on) of a step: update_k = a_k * score_k + b_k * z_k
        (reference utils/sampling.py:125-193)."""
        cfg, sg = self.cfg, self.cfg.sigma
        steps = len(schedule)
        t = float(schedule[t_idx])
        dt = float(schedule[t_idx] - schedule[t_idx + 1]) if t_idx < steps - 1 else float(schedule[t_idx])
        noise_off = cfg.no_random or (cfg.no_final_step_noise and t_idx == steps - 1)
        coef = []
        for k, (lo, hi, two) in enumerate(((sg.tr_sigma_min, sg.tr_sigma_max, True), (sg.rot_sigma_min, sg.rot_sigma_max, False),
                                           (sg.tor_sigma_min, sg.tor_sigma_max, True),
                                           (sg.sidechain_tor_sigma_min, sg.sidechain_tor_sigma_max, True))):
            sigma = lo ** (1 - t) * hi ** t
            g = sigma * math.sqrt(2 * math.log(hi / lo)) if two else 2 * sigma * math.sqrt(math.log(hi / lo))
            if cfg.ode:
                a, b = 0.5 * g ** 2 * dt, 0.0
            elif cfg.temp_sampling[k] != 1.0:
                sigma_data = math.exp(cfg.temp_sigma_data * math.log(hi) + (1 - cfg.temp_sigma_data) * math.log(lo))
                lam = (sigma_data + sigma) / (sigma_data + sigma / cfg.temp_sampling[k])
                a, b = g ** 2 * dt * (lam + cfg.temp_sampling[k] * cfg.temp_psi[k] / 2), g * math.sqrt(dt * (1 + cfg.temp_psi[k]))
            else:
                a, b = g ** 2 * dt, g * math.sqrt(dt)
            coef += [a, b]
        return t, coef, noise_off

    def _upload_step(self, t_idx, schedule):
        """Time, coefficients and noise of the step -> the device buffer: one asynchronous copy from a pinned row (_pinned_row; a
        copy from pageable memory would make the host wait for the stream to drain).  Noise is drawn for ALL samples of the job from the seeded generator - in the
        reference loop's order tr, rot, tor, side chains - and sliced to this shard."""
        cfg, N, sl, n = self.cfg, self.n_total, self.slice, self.n
        t, coef, noise_off = self._step_coefficients(t_idx, schedule)
        host = self._pinned_row()
        host[0] = t
        host[1:8] = 0.0
        host[8:16] = torch.tensor(coef, dtype=torch.float64).float()
        o = self._off

        def z(shape, key, cols):
            full = torch.zeros(shape) if noise_off else torch.randn(shape, generator=self.gen)
            host[o[key]:o[key] + n * cols] = full[sl].reshape(-1)

        z((N, 3), "z_tr", 3)
        z((N, 3), "z_rot", 3)
        if self.z_tor is not None:
            z((N, self.T), "z_tor", self.T)
        if self.z_sc is not None:
            z((N, self.S), "z_sc", self.S)
        self.params.copy_(host, non_blocking=True)
        self._pin_ev[self._pin_k].record()

    def _pinned_row(self):
        """The pinned block the step's parameters travel from: a ring of rows allocated ONCE (a fresh `torch.empty(pin_memory=True)`
        per step finds every earlier block still in flight while the host runs ahead of the device, so steps kept paying a
        hipHostMalloc - usually ~30 us, but single calls of 1 and of 84 ms were measured inside 20-step jobs, the latter +4.2 ms per
        step of a cfg1 job: section 8 of DESIGN.md).  A row is reused once
        the copy issued from it has run (its event; never waited for in practice: 64 steps of run-ahead)."""
        if self._pin is None:
            rows = 64
            self._pin = torch.empty((rows, self._n_par), pin_memory=True)
            self._pin_ev = [torch.cuda.Event() for _ in range(rows)]
            self._pin_used = [False] * rows
            self._pin_k = -1
        k = self._pin_k = (self._pin_k + 1) % len(self._pin_ev)
        if self._pin_used[k]:
            self._pin_ev[k].synchronize()
        self._pin_used[k] = True
        return self._pin[k]

    def _step_body(self):
        """Everything of a step that runs on the device: score model, SDE step, side-chain and ligand pose update (in place)."""
        from . import _lib as L
        import ctypes as C
        lib = L.load()
        cfg = self.cfg
        tr_score, rot_score, tor_score, sc_score = self._call_model(self.model, self.batch)
        a = L.SdeArgs()
        use_tor = self.z_tor is not None
        comps = [(tr_score, self.z_tr, self.upd["tr"]), (rot_score, self.z_rot, self.upd["rot"]),
                 (tor_score if use_tor else None, self.z_tor, self.upd["tor"]), (sc_score if self.has_flex else None, self.z_sc, self.upd["sc"])]
        keep = []
        for k, (sc, zz, out) in enumerate(comps):
            if sc is None or out is None or out.numel() == 0:
                a.n[k] = 0
                continue
            sc = sc.contiguous()
            keep.append(sc)
            a.score[k], a.z[k], a.out[k], a.n[k] = sc.data_ptr(), (0 if cfg.ode else zz.data_ptr()), out.data_ptr(), out.numel()
        st = torch._C._cuda_getCurrentRawStream(self.lig_pos.device.index)
        L.check(lib.ddp_sde_update(self.coef.data_ptr(), C.byref(a), st), "ddp_sde_update")
        if self.has_flex:
            L.check(lib.ddp_sidechain_update(self.atom_pos.data_ptr(), self.n, self.n_a, self.upd["sc"].data_ptr(), self.S,
                                             self.sc_edge_i32.data_ptr(), self.sc_sub_i32.data_ptr(), self.sc_map_i32.data_ptr(),
                                             self.atom_pos.data_ptr(), st), "ddp_sidechain_update")
            torch.autograd.graph.increment_version(self.atom_pos)     # moved in place: the model's static-graph cache keys on it
        L.check(lib.ddp_pose_update(self.lig_pos.data_ptr(), self.n, self.n_l, self.upd["tr"].data_ptr(), self.upd["rot"].data_ptr(),
                                    self.upd["tor"].data_ptr() if use_tor else None, self.T if use_tor else 0,
                                    self.bonds_i32.data_ptr() if use_tor else None, self.mask_u8.data_ptr() if use_tor else None,
                                    self.lig_pos.data_ptr(), st), "ddp_pose_update")
        torch.autograd.graph.increment_version(self.lig_pos)

    def _step_hip(self, t_idx, schedule):
        with torch.cuda.device(self.device):
            self._bind_batch()
            self._upload_step(t_idx, schedule)
            if self._graph and self._graph_epoch != getattr(self.model, "_packed_epoch", 0):
                # the model dropped its packed weights / static caches (model.to(), load_state_dict, changed weights found by
                # _refresh_weight_caches): the captured launches read memory that is no longer held - recapture from new state
                self._graph, self._graph_keep, self._steps_run = None, None, 0
            if self._graph and self.graph_enabled:
                self._graph.replay()
                if self._graph_stats is not None:     # (the replay rewrote the count block the captured forward's stats read)
                    self.model.last_stats = self._graph_stats.fresh()
                # (the replay moved the poses in place behind Python's back: the model's static-graph cache keys on the versions)
                torch.autograd.graph.increment_version(self.lig_pos)
                if self.has_flex:
                    torch.autograd.graph.increment_version(self.atom_pos)
            elif self.cfg.hip_graph and self.graph_enabled and self._steps_run >= 2 and self._graph is None:
                self._capture()
            else:
                self._step_body()
            self._steps_run += 1

    def _capture(self):
        """Capture one step in a hipGraph (the step being captured is also executed: capture records, then the graph is
        replayed once).  Anything that cannot be captured makes the sampler fall back to ordinary launches for good."""
        g = torch.cuda.CUDAGraph()
        # Whatever the model's static-graph cache holds for the CURRENT poses (a forward on them may just have run: Sampler.scores)
        # must be recomputed INSIDE the graph, not referenced by it: the cache keys on the pose tensors' version counters
        torch.autograd.graph.increment_version(self.lig_pos)
        if self.has_flex:     # (a rigid receptor's entries are genuinely static - and some of them synchronise when they are built)
            torch.autograd.graph.increment_version(self.atom_pos)
        try:
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(g, pool=self._graph_pool()):
                self._step_body()
        except Exception as e:     # noqa: BLE001
            import warnings
            warnings.warn(f"hipGraph capture of the denoising step failed ({type(e).__name__}: {e}); running it launch by launch",
                          RuntimeWarning)
            self._graph = False
            torch.cuda.synchronize(self.device)
            # what the aborted capture left in the model's static-graph cache was never computed
            for mdl in (self.model,):
                if hasattr(mdl, "_static_cache"):
                    mdl._static_cache = {}
            self._step_body()
            return
        self._graph = g
        self._graph_epoch = getattr(self.model, "_packed_epoch", 0)
        st = getattr(self.model, "last_stats", None)
        self._graph_stats = st if hasattr(st, "fresh") else None
        # Everything the capture left in the model's static-graph cache lives in the graph's memory pool and is read and
        # written by its replays: those tensors must not be freed while the graph is alive, whatever later forwards put in
        # the cache's slots
        b = self.batch
        self._graph_keep = [[dict(c) for c in self.model.__dict__.get("_static_caches", {}).values()],
                            getattr(self.model, "last_stats", None), getattr(b, "graph_sigma_emb", None),
                            [getattr(b[nt], "node_sigma_emb", None) for nt in ("ligand", "receptor", "atom")],
                            getattr(b["atom", "atom"], "edge_index", None),     # (what the captured forward left on model and batch)
                            self._packed_refs()]
        g.replay()

    # Memory pool of the captured steps.  A captured 40-sample step keeps its intermediates (25 - 45 GB: the G arrays of every layer)
    # in the graph's memory pool.  With a pool per graph PyTorch hands the segments back (hipFree) when the graph dies - and on this
    # ROCm stack memory allocated during stream capture is NOT returned to the device by that (measured: tests that build a model and
    # a captured 40-sample sampler each left 35 - 45 GB behind with torch.cuda.memory_reserved() at 0.1 GB, until the device's 288 GB
    # were gone; a csv run over a few complexes would end the same way).  So the captured steps of a process share ONE pool per
    # device: the next capture reuses the segments of the previous, dead one.  PyTorch's rule for shared pools - replay in capture
    # order, one at a time - holds when at most one captured step is alive; a sampler that captures while another captured sampler
    # of the device is still alive gets a pool of its own.
    _pools = {}          # device index -> (torch.cuda.MemPool, [weak references to the samplers that captured into it])

    def _graph_pool(self):
        import weakref
        idx = torch.device(self.device).index or 0
        ent = Sampler._pools.get(idx)
        if ent is None:
            # (a MemPool OBJECT, kept for the life of the process: while it lives the pool's use count stays above zero, so neither a
            # dying graph nor the torch.cuda.empty_cache() that torch.cuda.graph runs before every capture hands its segments back)
            with torch.cuda.device(idx):
                ent = Sampler._pools[idx] = (torch.cuda.MemPool(), [])
        live = [r for r in ent[1] if r() is not None and r() is not self and r()._graph]
        ent[1][:] = live
        if live:
            return None
        ent[1].append(weakref.ref(self))
        return ent[0].id

    def close(self):
        """Drops the captured step and everything it keeps alive (the graph's private memory pool holds a step's intermediates: ~25 GB
        at 40 samples).  The sampler goes on launch by launch if it is stepped again."""
        g = self._graph
        self._graph, self._graph_keep, self._graph_stats = False, None, None
        if g:
            g.reset()

    def _packed_refs(self):
        """Packed weights, edge-MLP packs and stage-A stacks built by the steps before the capture live OUTSIDE the graph's memory
        pool while the captured launches carry their addresses: the graph holds them, so that `model.invalidate_packed()` (which
        only drops the model's references) can never hand that memory to someone else under a graph that is still replayed."""
        m = self.model
        refs = [dict(getattr(m, "_edge_packs", {}) or {}), dict(getattr(m, "_stage_a_stacks", {}) or {})]
        if hasattr(m, "modules"):
            refs += [(getattr(c, "_packed", None), getattr(c, "_packed_g", None)) for c in m.modules() if hasattr(c, "_packed")]
        return refs

    # -- randomize_position (reference utils/sampling.py:16-60), pocket_knowledge=False -----------------------
    def randomize(self):
        cfg, N, sl = self.cfg, self.n_total, self.slice
        if not cfg.no_torsion and self.T > 0:
            ang = (torch.rand((N, self.T), generator=self.gen) * 2 - 1) * math.pi
            self._set_pos("lig_pos", apply_torsions(self.lig_pos, self.bonds, self.rot_idx, ang[sl].to(self.device)))
        if self.has_flex:
            ang = (torch.rand((N, self.S), generator=self.gen) * 2 - 1) * math.pi
            self._set_pos("atom_pos", apply_sidechain_torsions(self.atom_pos, self.sc_edge_idx, self.sc_sub, self.sc_map,
                                                               ang[sl].to(self.device)))
        # uniform random rotations from normalised gaussian quaternions (scipy Rotation.random)
        q = torch.randn((N, 4), generator=self.gen)
        q = (q / q.norm(dim=1, keepdim=True))[sl].to(self.device)
        w, x, y, z = q.unbind(1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
        center = self.lig_pos.mean(1, keepdim=True)
        new = (self.lig_pos - center) @ R.transpose(1, 2)
        if not cfg.no_random:
            tr = torch.randn((N, 1, 3), generator=self.gen) * cfg.sigma.tr_sigma_max
            new = new + tr[sl].to(self.device)
        self._set_pos("lig_pos", new)

    def _set_pos(self, name, value):
        """Poses are replaced on the CPU path and written IN PLACE on a HIP device (the batch holds views of them and a captured
        step their addresses)."""
        if self.on_hip:
            getattr(self, name).copy_(value)
        else:
            setattr(self, name, value)

    # -- one denoising step (reference utils/sampling.py:93-251) ------------------------------------------------
    def step(self, t_idx: int, schedule: np.ndarray):
        if self.on_hip:
            return self._step_hip(t_idx, schedule)
        cfg, sg, N, sl, dev = self.cfg, self.cfg.sigma, self.n_total, self.slice, self.device
        steps = len(schedule)
        t = float(schedule[t_idx])
        dt = float(schedule[t_idx] - schedule[t_idx + 1]) if t_idx < steps - 1 else float(schedule[t_idx])
        tr_s = sg.tr_sigma_min ** (1 - t) * sg.tr_sigma_max ** t
        rot_s = sg.rot_sigma_min ** (1 - t) * sg.rot_sigma_max ** t
        tor_s = sg.tor_sigma_min ** (1 - t) * sg.tor_sigma_max ** t
        sc_s = sg.sidechain_tor_sigma_min ** (1 - t) * sg.sidechain_tor_sigma_max ** t

        noise_off = cfg.no_random or (cfg.no_final_step_noise and t_idx == steps - 1)

        def z(shape):  # drawn for all samples of the job, sliced to this shard
            full = torch.zeros(shape) if noise_off else torch.randn(shape, generator=self.gen)
            part = full[sl]
            if torch.device(dev).type == "cuda":
                # pinned staging + asynchronous copy: a copy from pageable memory makes the host wait until the stream has
                # drained, i.e. for the previous step's conv layers - and everything the host could have queued behind them
                # (time tensors, node encoders, the neighbour-search count passes) would start late
                return part.contiguous().pin_memory().to(dev, non_blocking=True)
            return part.to(dev)

        # the step's noise does not depend on the scores: drawn (same generator order as the reference loop: tr, rot, tor,
        # side chains) and uploaded before the model call, so that nothing between the model and the pose update waits
        z_tr, z_rot = z((N, 3)), z((N, 3))
        z_tor = z((N, self.T)) if (not cfg.no_torsion and self.T > 0) else None
        z_sc = z((N, self.S)) if self.has_flex else None

        b = self.batch
        b["ligand"].pos = self.lig_pos.reshape(-1, 3)
        b["atom"].pos = self.atom_pos.reshape(-1, 3)
        set_time(b, t, t, t, t, device=dev)
        tr_score, rot_score, tor_score, sc_score = self._call_model(self.model, b)

        def perturb(score, g, sigma, lo, hi, k, zz):
            if cfg.ode:
                return 0.5 * g ** 2 * dt * score
            if cfg.temp_sampling[k] != 1.0:
                sigma_data = math.exp(cfg.temp_sigma_data * math.log(hi) + (1 - cfg.temp_sigma_data) * math.log(lo))
                lam = (sigma_data + sigma) / (sigma_data + sigma / cfg.temp_sampling[k])
                return (g ** 2 * dt * (lam + cfg.temp_sampling[k] * cfg.temp_psi[k] / 2) * score
                        + g * math.sqrt(dt * (1 + cfg.temp_psi[k])) * zz)
            return g ** 2 * dt * score + g * math.sqrt(dt) * zz

        tr_g = tr_s * math.sqrt(2 * math.log(sg.tr_sigma_max / sg.tr_sigma_min))
        rot_g = 2 * rot_s * math.sqrt(math.log(sg.rot_sigma_max / sg.rot_sigma_min))
        tr_p = perturb(tr_score, tr_g, tr_s, sg.tr_sigma_min, sg.tr_sigma_max, 0, z_tr)
        rot_p = perturb(rot_score, rot_g, rot_s, sg.rot_sigma_min, sg.rot_sigma_max, 1, z_rot)
        tor_p = None
        if not cfg.no_torsion and self.T > 0:
            tor_g = tor_s * math.sqrt(2 * math.log(sg.tor_sigma_max / sg.tor_sigma_min))
            tor_p = perturb(tor_score.reshape(self.n, self.T), tor_g, tor_s, sg.tor_sigma_min, sg.tor_sigma_max, 2, z_tor)
        if self.has_flex:
            sc_g = sc_s * math.sqrt(2 * math.log(sg.sidechain_tor_sigma_max / sg.sidechain_tor_sigma_min))
            sc_p = perturb(sc_score.reshape(self.n, self.S), sc_g, sc_s, sg.sidechain_tor_sigma_min,
                           sg.sidechain_tor_sigma_max, 3, z_sc)
            if self.atom_pos.is_cuda:
                self.atom_pos = apply_sidechain_torsions_hip(self.atom_pos, self.sc_edge_i32, self.sc_sub_i32, self.sc_map_i32, sc_p)
            else:
                self.atom_pos = apply_sidechain_torsions(self.atom_pos, self.sc_edge_idx, self.sc_sub, self.sc_map, sc_p)
        if self.lig_pos.is_cuda:   # one HIP launch; the PyTorch form below is the same arithmetic (CPU tests)
            self.lig_pos = modify_conformer_hip(self.lig_pos, tr_p, rot_p, tor_p, self.bonds_i32, self.mask_u8)
        else:
            self.lig_pos = modify_conformer(self.lig_pos, tr_p, rot_p, tor_p, self.bonds, self.rot_idx)

    def _call_model(self, model, b):
        """The weights' VALUE fingerprint (score_model._refresh_weight_caches: one host synchronisation) is checked on the
        first call of a run only; the remaining steps trust the version counters."""
        seen = self.__dict__.setdefault("_weights_checked", set())
        if not hasattr(model, "check_weight_values") or id(model) not in seen:
            seen.add(id(model))
            return self._model_call(model, b)
        prev, model.check_weight_values = model.check_weight_values, False
        try:
            return self._model_call(model, b)
        finally:
            model.check_weight_values = prev

    def _model_call(self, model, b):
        """model(b) with the per-forward h2 range check (a host synchronisation) switched off: a sampler checks once per run and recovers
        the whole run (run / confidence)."""
        prev = getattr(model, "range_check_in_forward", None)
        if prev is None:
            return model(b)
        model.range_check_in_forward = False
        try:
            return model(b)
        finally:
            model.range_check_in_forward = prev

    # -- confidence pass + ranking (reference utils/sampling.py:263-283, inference.py:212-219) ------------------------
    def confidence(self, confidence_model):
        """Runs the confidence model (TensorProductScoreModel(confidence_mode=True)) on the final poses at t = 0 and
        returns (confidence [n] or [n, k], order) with `order` = sample indices from most to least confident (for a
        multi-output head the reference ranks by the first column, inference.py:213-216)."""
        b = self.batch
        if self.on_hip:     # the batch's poses and times are views of the sampler's device buffers
            self._bind_batch()
            self.t_dev.zero_()
        else:
            b["ligand"].pos = self.lig_pos.reshape(-1, 3)
            b["atom"].pos = self.atom_pos.reshape(-1, 3)
            set_time(b, 0.0, 0.0, 0.0, 0.0, device=self.device)
        conf = self._call_model(confidence_model, b)
        if self.on_hip and hasattr(confidence_model, "check_overflow"):
            # a truncated ligand<-atom list is reported for THIS complex, before its scores are ranked (the flag is written by the
            # search kernel: visible once the forward has run); a value outside the fp16 range reruns the pass in the fp32 form
            torch.cuda.synchronize(self.device)
            if hasattr(confidence_model, "range_flag_raised") and confidence_model.range_flag_raised():
                conf = confidence_model.forward_fp32(b)
                torch.cuda.synchronize(self.device)
            confidence_model.check_overflow()
        key = conf[:, 0] if conf.dim() == 2 else conf
        return conf, torch.argsort(key, descending=True)

    def run(self, schedule: Optional[np.ndarray] = None):
        schedule = get_t_schedule(self.cfg.inference_steps) if schedule is None else schedule
        self.__dict__["_weights_checked"] = set()      # a run re-checks the weights' values once
        if self.on_hip and self._graph and hasattr(self.model, "_refresh_weight_caches"):
            # replays never enter the model's Python forward: the value check (param.data.copy_, EMA) is made here, and a change
            # drops the packed weights -> the epoch test in _step_hip recaptures
            self.model._refresh_weight_caches()
        snap = self.snapshot() if self.on_hip else None
        for i in range(len(schedule)):
            self.step(i, schedule)
        if self.on_hip and hasattr(self.model, "range_flag_raised"):
            # The fp16 hi/lo form of the fc products is not total (|v| > 65504 cannot be split); its kernels report such a value through a
            # flag in pinned host memory.  One check per run: a raised flag puts the job back to its first step (poses, noise stream) and
            # runs it again in the exact fp32 MFMA form - a new capture, the model's switch restored afterwards.
            torch.cuda.synchronize(self.device)
            if self.model.range_flag_raised():
                self.model.__dict__["h2_recoveries"] = self.model.__dict__.get("h2_recoveries", 0) + 1
                prev = self.model.conv_h2
                self.model.conv_h2 = False          # (bumps the packed epoch: the captured step is dropped and recaptured)
                try:
                    self.restore(snap)
                    for i in range(len(schedule)):
                        self.step(i, schedule)
                    torch.cuda.synchronize(self.device)
                finally:
                    self.model.conv_h2 = prev
        self.check_overflow()
        return self.lig_pos, self.atom_pos

    def check_overflow(self):
        """Raises if the ligand<-atom list of a step queued so far was truncated (model.la_capacity_per_atom).  Inside a replayed
        step the model's Python forward - which checks at its top - is never entered: callers that drive `step` themselves
        (bench.py, tools) call this after their last step; `run` and `confidence` do."""
        if self.on_hip and hasattr(self.model, "check_overflow"):
            torch.cuda.synchronize(self.device)
            self.model.check_overflow()


class PipelinedSampler:
    """The same job as `Sampler`, with the local samples split into `ways` contiguous groups that are stepped one after
    the other on their own HIP streams.

    Why: the front of a forward (neighbour searches, CSR views, head graphs) is a chain of small launches with a few
    host synchronisations - the host, not the device, sets its pace - while the conv layers that follow keep the device
    busy for tens of milliseconds with the host far ahead.  With two resident groups the host prepares and queues group
    B's front while the device still runs group A's conv layers (and the other way round), so the device does not wait
    for the host between steps.  Every group is a `Sampler` over its own `sample_slice` of the same seeded job (noise is
    drawn for the whole job and sliced): the poses are bit for bit those of the groups run on their own
    (`tests/test_gpu_parity.py::test_sampler_end_to_end_on_device`), i.e. those of the single-batch `Sampler` up to the
    fp32 rounding of a few batch-level reductions, exactly like sharding over GPUs.
    Each group keeps its own slot of the model's static-graph cache (`model.cache_slot`)."""

    def __init__(self, model, complex_graph: HeteroBatch, n_total: int, device, cfg: SamplerConfig, seed: int = 0,
                 sample_slice: Optional[slice] = None, ways: int = 2):
        self.model, self.device = model, device
        lo, hi, _ = (sample_slice or slice(0, n_total)).indices(n_total)
        ways = max(1, min(ways, hi - lo))
        import dataclasses
        cfg = dataclasses.replace(cfg, hip_graph=False)      # (the groups switch streams inside a step: launched one by one)
        cuts = [lo + (hi - lo) * w // ways for w in range(ways + 1)]
        self.streams = [torch.cuda.Stream(device=device, priority=-1) for _ in range(ways)]       # fronts: high priority
        self.layer_streams = [torch.cuda.Stream(device=device, priority=0) for _ in range(ways)]
        self._part_done = [None] * ways
        self.parts: List[Sampler] = []
        torch.cuda.synchronize(device)
        for st, a, b in zip(self.streams, cuts[:-1], cuts[1:]):
            with torch.cuda.stream(st):
                self.parts.append(Sampler(model, complex_graph, n_total, device, cfg, seed=seed, sample_slice=slice(a, b)))
        torch.cuda.synchronize(device)
        self.n = hi - lo
        self.n_l, self.n_a = self.parts[0].n_l, self.parts[0].n_a
        self.on_hip = torch.device(device).type == "cuda"
        self._stats: list = []
        self._warm = False
        self._done = None          # event after the most recent group step

    def snapshot(self):
        """One `Sampler.snapshot()` per group (poses, generator state), taken with the device idle."""
        torch.cuda.synchronize(self.device)
        return [p.snapshot() for p in self.parts]

    def restore(self, snap):
        """Puts every group back to its snapshot (on the group's own stream, behind whatever its last step queued)."""
        snaps = iter(snap)
        self._each(lambda p: p.restore(next(snaps)))
        torch.cuda.synchronize(self.device)

    def _each(self, fn):
        stats = []
        for slot, part in enumerate(self.parts):
            F, Ls = self.streams[slot], self.layer_streams[slot]
            self.model.cache_slot = slot

            def to_layers(F=F, Ls=Ls):
                # front -> layers handoff: the layers go to the group's normal-priority stream, after this group's front
                # and after the previous group's step.  The conv layers of the groups run one after the other - two
                # saturating kernel sequences sharing the CUs would both finish late and the host could not start either
                # group's next front early; only a group's front (high-priority stream, small kernels) overlaps them.
                ev = torch.cuda.Event()
                ev.record(F)
                Ls.wait_event(ev)
                if self._done is not None:
                    Ls.wait_event(self._done)
                torch.cuda.set_stream(Ls)

            self.model.before_layers = to_layers
            try:
                with torch.cuda.stream(F):
                    if self._part_done[slot] is not None:    # the front reads what the group's last step left on Ls
                        F.wait_event(self._part_done[slot])
                    fn(part)
                    done = torch.cuda.Event()
                    done.record(torch.cuda.current_stream())
                    self._done = self._part_done[slot] = done
            finally:
                self.model.before_layers = None
            stats.append(self.model.last_stats)      # (read lazily: the edge counts live on the device)
            if not self._warm:   # weights are packed on first use: finished before another stream reads them
                torch.cuda.synchronize(self.device)
        self.model.cache_slot = 0
        self._warm = True
        return stats

    def randomize(self):
        self._each(lambda p: p.randomize())

    def step(self, t_idx: int, schedule: np.ndarray):
        self._stats = self._each(lambda p: p.step(t_idx, schedule))

    @property
    def last_stats(self):
        """Edge / node / graph counts of the last step, summed over the groups."""
        stats = [dict(s) for s in self._stats]
        return {k: sum(s.get(k, 0) for s in stats) for k in stats[0]} if stats else {}

    def _gather(self, name):
        torch.cuda.synchronize(self.device)
        return torch.cat([getattr(p, name) for p in self.parts], 0)

    @property
    def lig_pos(self):
        return self._gather("lig_pos")

    @property
    def atom_pos(self):
        return self._gather("atom_pos")

    def run(self, schedule: Optional[np.ndarray] = None):
        schedule = get_t_schedule(self.parts[0].cfg.inference_steps) if schedule is None else schedule
        snap = self.snapshot() if self.on_hip else None
        for i in range(len(schedule)):
            self.step(i, schedule)
        if self.on_hip and hasattr(self.model, "range_flag_raised"):
            # The fp16 hi/lo form of the fc products is not total (|v| > 65504 cannot be split); its kernels report such a value through a
            # flag in pinned host memory.  One check per run: a raised flag puts the job back to its first step (poses, noise stream) and
            # runs it again in the exact fp32 MFMA form - a new capture, the model's switch restored afterwards.
            torch.cuda.synchronize(self.device)
            if self.model.range_flag_raised():
                self.model.__dict__["h2_recoveries"] = self.model.__dict__.get("h2_recoveries", 0) + 1
                prev = self.model.conv_h2
                self.model.conv_h2 = False          # (bumps the packed epoch: the captured step is dropped and recaptured)
                try:
                    self.restore(snap)
                    for i in range(len(schedule)):
                        self.step(i, schedule)
                    torch.cuda.synchronize(self.device)
                finally:
                    self.model.conv_h2 = prev
        self.check_overflow()
        return self.lig_pos, self.atom_pos

    def check_overflow(self):
        """Raises if a ligand<-atom list of any group's steps was truncated (see Sampler.check_overflow)."""
        if hasattr(self.model, "check_overflow"):
            torch.cuda.synchronize(self.device)
            self.model.check_overflow()
