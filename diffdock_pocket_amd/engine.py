"""The forward of the MI355X score model as a DEVICE-DRIVEN sequence of launches (reference models/all_atom_score_model.py:
238-436, called once per denoising step from utils/sampling.py:119-120).

Nothing in a steady-state step waits for the device: the sizes of the pose-dependent lists - the three radius graphs of
build_lig / build_cross_conv_graph (:444-583), the heads' bond-centre graphs (:586-636), their CSR / source-order views and the
sub-lists of the exact work eliminations - stay in device memory (launch.CountBlock); every consumer is launched on a grid
sized for the list's CAPACITY (worst case from the graph sizes) and reads the actual count itself (include/ddp_hip.h,
"Device-side counts").  What does synchronise, once per batch and cached (`model._cached`): the dense layouts of the batch
vectors, the exact comparison that finds the receptor side identical across the samples, and - only when the caller did not
say so through `batch.set_time` - the check that all receptor-side nodes sit at one diffusion time.

    front()   node encoders + sigma tables (1 launch), neighbour searches (3), edge embeddings (6), CSR / source-order views (10)
    lists()   the index lists of the exact eliminations: atoms a ligand message reaches, dead-output pruning of the last
              receptor-side layers, layer-1 clean-pair sharing - all as batched device list primitives (csrc/ddp_lists.hip)
    layers()  per layer: stage A of the factorised convs, the two conv launches, the segmented means
    heads()   tr / rot head, torsion heads, confidence head
"""
from __future__ import annotations

import ctypes as C
from collections.abc import Mapping
from types import SimpleNamespace
from typing import Dict, List, Optional

import math

import torch

from . import _lib as L
from . import graph as G
from . import launch as K
from . import packing as P
from . import packing as P_
from .graph import EdgeView

SRC_TYPE = {0: "l", 1: "r", 2: "a", 3: "a", 4: "l", 5: "r", 6: "r", 7: "l", 8: "a"}
RECV_TYPE = {0: "l", 1: "l", 2: "l", 3: "a", 4: "a", 5: "a", 6: "r", 7: "r", 8: "r"}
RECV_OF = {"a": (3, 4, 5), "r": (6, 7, 8)}
# summation order of the residual update (:316,:320,:324): lig u0+u2+u1, atom u3+u4+u5, rec u6+u8+u7
ORDER = {"l": [0, 2, 1], "a": [3, 4, 5], "r": [6, 8, 7]}
_DROP_SELF = 1


class LazyStats(Mapping):
    """Edge / node counts of the last forward.  The pose-dependent ones live on the device; they are copied to the host when
    the mapping is first read (one synchronisation, outside the step)."""

    def __init__(self, static: Dict[str, int], counts: Optional[K.CountBlock], names: Dict[str, str]):
        self._static, self._counts, self._names, self._vals = dict(static), counts, dict(names), None

    def _resolve(self):
        if self._vals is None:
            vals = dict(self._static)
            if self._counts is not None:
                host = self._counts.values()
                for key, slot in self._names.items():
                    if slot in host:
                        vals[key] = host[slot]
            self._vals = vals
        return self._vals

    def fresh(self):
        """A new, unread view over the same count block.  The block of a CAPTURED forward is rewritten by every replay of the
        step's hipGraph, while this object memoises its first read: sampler.Sampler installs a fresh view after each replay."""
        return LazyStats(self._static, self._counts, self._names)

    def __getitem__(self, k):
        return self._resolve()[k]

    def __iter__(self):
        return iter(self._resolve())

    def __len__(self):
        return len(self._resolve())


def _i32(t):
    return t.to(torch.int32).contiguous()


class _Fork:
    """Independent launches of one layer on side streams (fork / join with events; inside a captured step they become
    parallel branches of the hipGraph).  Used for SMALL batches only: a layer's direct-conv launch (receptor<-atom: one 64-edge
    workgroup per 64 atoms, ~0.3 ms each whatever the batch) and its stage-A products then overlap with the 32-edge conv
    kernel instead of running behind it - at 5 samples of 3dpf the kernels of a layer do not fill the chip one by one; at
    40 samples they do, and sharing the CUs was measured without gain."""

    def __init__(self, dev, n):
        self.main = torch.cuda.current_stream(dev)
        # (all at the default priority: a high-priority branch - the direct conv, or the conv launches - cost 1.9 - 2.2 ms of a 16-ms step,
        # profiles/r06_direct_conv_priority.txt)
        self.side = [torch.cuda.Stream(device=dev) for _ in range(n)]
        self.used = []

    def run(self, i, fn):
        st = self.side[i % len(self.side)]
        if st not in self.used:
            st.wait_stream(self.main)     # fork: after everything queued so far
            self.used.append(st)
        with torch.cuda.stream(st):
            return fn()

    def resync(self, i):
        """A forked slot additionally waits for what the main stream has queued since the fork."""
        st = self.side[i % len(self.side)]
        if st in self.used:
            st.wait_stream(self.main)

    def join(self, only=None):
        """Main waits for the forked streams (`only`: for that slot alone, the others stay forked)."""
        for st in list(self.used):
            if only is None or st is self.side[only % len(self.side)]:
                self.main.wait_stream(st)
                self.used.remove(st)


class ForwardEngine:
    def __init__(self, model):
        self.m = model
        self._forks = {}

    def _fork(self, dev):
        """Side streams of the current stream for a layer's independent launches."""
        f = self._forks.get(dev)     # (created by the first ordinary step: no stream is created during a capture)
        if f is None:
            # slots 0 - 2: stage-A groups / front, 3: direct conv, 4: index lists (pipelined order: 0 = the early conv launch, 1 = the direct
            # conv, 2 / 3 = the ligand / receptor chains)
            f = self._forks[dev] = _Fork(dev, 5)
        f.main = torch.cuda.current_stream(dev)
        return f

    # ================================================================================================ entry
    @torch.no_grad()
    def forward(self, data):
        lig = data["ligand"]
        K.require_hip(lig.pos)
        with torch.cuda.device(lig.pos.device):      # every launch below goes to the current stream of THIS device
            return self._forward(data)

    def _forward(self, data):
        m = self.m
        lig, rec, atom = data["ligand"], data["receptor"], data["atom"]
        dev = lig.pos.device
        m._refresh_weight_caches()
        m.rows_all_or_none(dev)
        m.check_overflow(range_too=m.range_check_in_forward)
        K.CONV_H2 = m.conv_h2       # the form of the fc products this forward's tasks are built for (score_model.conv_h2)
        if dev in self._forks:      # (a forward that raised may have left streams marked as forked: the next fork waits for main again)
            self._forks[dev].used.clear()
        K.set_range_flag(m.overflow_flag(dev)[1:2])     # where this forward's h2 kernels report values outside the fp16 range
        mark = m.section_timer.mark if m.section_timer is not None else (lambda name: None)
        mark("start")
        if m.no_aminoacid_identities:
            rec.x = rec.x * 0
        S = self._static(data, lig, rec, atom, dev)
        # the front's independent chains on forked streams (model.fork_front); not under the section timer / hooks / debug outputs
        ff = None
        if (m.fork_front and m.section_timer is None and m.before_layers is None and m.debug_conv_outputs is None
                and not m.exact_sizes):
            ff = self._fork(dev)
        F = self._front(data, S, lig, rec, atom, dev, mark, ff)
        # The index lists read graph structure only.  Rigid receptor: layer 1 is their first reader, so they run on a stream of
        # their own beside stage A and the 32-edge conv launch of layer 0 (_layers joins them); with flexible side chains layer 0
        # reads the flex0 lists, which stay on the main stream with the layer-1 lists that build on them - the pruned lists of the
        # last layers still go to the side stream (_lists).  (Launch by launch in round 3 this was measured without gain - 33.0 ms either way, cfg1 x 4
        # samples 1.6 -> 1.9 ms of host time; as a branch of the captured step it is free.)
        F.lists_fork = ff
        self._lists(S, F, dev, side=(lambda fn: ff.run(4, fn)) if ff is not None else None)
        mark("lists")
        if m.exact_sizes:
            self._exact(F)
        if m.before_layers is not None:
            m.before_layers()
        self._layers(S, F, dev, mark)
        out = self._heads(data, S, F, lig, rec, atom, dev, mark)
        m.last_stats = LazyStats(S.stats, F.cnt, {"E_ll": "ll", "E_lr": "lr", "E_la": "la", "n_near": "near",
                                                  "clean1_dirty_edges": "dirty", "flex0_kept_aa_edges": "fx_3"})
        prof = K.profiler()
        if prof is not None:
            prof.hold = getattr(prof, "hold", [])
            prof.hold.append(F.cnt.block)
        return out

    # ================================================================================================ static state
    def _static(self, data, lig, rec, atom, dev):
        """Everything that does not change between the denoising steps of one batch (kept by `model._cached` while the tensors
        it was derived from are the same objects with the same version counters)."""
        m = self.m
        S = SimpleNamespace()
        B = S.B = int(data.num_graphs)
        lbatch, rbatch, abatch = lig.batch.long(), rec.batch.long(), atom.batch.long()
        S.lbatch, S.rbatch, S.abatch = lbatch, rbatch, abatch
        S.Nl, S.Nr, S.Na = lig.pos.shape[0], rec.pos.shape[0], atom.pos.shape[0]
        S.lay_l = m._cached("lay_l", (lig.batch,), lambda: G.DenseLayout.build(lbatch, B))
        S.lay_r = m._cached("lay_r", (rec.batch,), lambda: G.DenseLayout.build(rbatch, B))
        S.lay_a = m._cached("lay_a", (atom.batch,), lambda: G.DenseLayout.build(abatch, B))
        for lay in (S.lay_l, S.lay_r, S.lay_a):
            G._ptr(lay)
        bond_ei = data["ligand", "ligand"].edge_index
        S.bond_ei = bond_ei
        S.bond32 = m._cached("bond32", (bond_ei,), lambda: (_i32(bond_ei[0]), _i32(bond_ei[1])))
        S.E_bond = int(bond_ei.shape[1])
        rr, ar = data["receptor", "receptor"].edge_index, data["atom", "receptor"].edge_index
        S.rr, S.ar = rr, ar
        S.num_flex = 0
        # (:327, literally: a PyG HeteroData answers `in` by attribute names, not node types, and creates the store on access)
        if m.flexible_sidechains and len(data["flexResidues"]) > 0:
            S.num_flex = int(data["flexResidues"].edge_idx.shape[0])
        nl, nr, na = S.lay_l.counts_host, S.lay_r.counts_host, S.lay_a.counts_host
        # worst-case sizes of the pose-dependent edge lists (per graph: every query x every point of its graph, or the cap)
        # (ligand radius graph: searched with 33 matches per query, then self is dropped - a query whose first 33 in-radius
        # matches in index order all precede it keeps 33 of them)
        S.cap_ll = S.E_bond + sum(n * min(max(n - 1, 0), 33) for n in nl)
        S.cap_lr = sum(a * min(b, 10000) for a, b in zip(nl, nr))
        S.cap_la = sum(a * min(b, 10000, max(int(m.la_capacity_per_atom), 1)) for a, b in zip(nl, na))
        S.stats = {"E_rr": int(rr.shape[1]), "E_ar": int(ar.shape[1]), "N_l": S.Nl, "N_r": S.Nr, "N_a": S.Na, "B": B}
        S.tor = S.sc = None
        if not m.confidence_mode:
            if not m.no_torsion:
                def tor_static():   # rotatable bonds, their graph index and dense layout: fixed for a batch
                    idx = lig.edge_mask.bool().nonzero(as_tuple=True)[0]
                    bnd = bond_ei[:, idx].long()
                    bb = lbatch[bnd[0]]
                    lay = G.DenseLayout.build(bb, B) if idx.shape[0] > 0 else None
                    return SimpleNamespace(idx=idx, bonds=bnd, batch=bb, lay=lay, T=int(idx.shape[0]), flat32=_i32(bnd.reshape(-1)),
                                           batch32=_i32(bb))
                tor = m._cached("tor_static", (lig.edge_mask, bond_ei, lig.batch), tor_static)
                if tor.T > 0:
                    G._ptr(tor.lay)
                    tor.cap = sum(t * min(n, 32) for t, n in zip(tor.lay.counts_host, nl))
                    S.tor = tor
            if S.num_flex > 0:
                fr = data["flexResidues"]

                def sc_static():
                    sb = fr.batch.long()
                    bonds = S.lay_a.starts[sb] + fr.edge_idx.t().long()     # get_sc_tor_bonds (:638-652)
                    return SimpleNamespace(bonds=bonds, batch=sb, lay=G.DenseLayout.build(sb, B), T=int(sb.shape[0]),
                                           flat32=_i32(bonds.reshape(-1)), batch32=_i32(sb))
                sc = m._cached("sc_static", (fr.batch, fr.edge_idx, atom.batch), sc_static)
                G._ptr(sc.lay)
                sc.cap = sum(t * min(n, 32) for t, n in zip(sc.lay.counts_host, na))
                S.sc = sc
        return S

    def _one_time(self, data, lig, rec, atom):
        """Do all receptor-side nodes sit at ONE diffusion time (the sampling batch)?  `batch.set_time` says so for the time
        tensors it made; for any others the device is asked (one host synchronisation)."""
        ts = (rec.node_t["tr"], atom.node_t["tr"])
        if all(t.numel() > 0 and (t.numel() == 1 or t.stride(0) == 0) for t in ts) and ts[0].data_ptr() == ts[1].data_ptr():
            return True      # stride-0 views of ONE device scalar (sampler.Sampler): one time by construction
        hint = getattr(data, "ddp_time_hint", None)
        if hint is not None and hint[0] == tuple(id(t) for t in ts) and hint[1] == tuple(t._version for t in ts):
            return bool(hint[2])
        t_nodes = torch.cat([t.reshape(-1) for t in ts])
        return bool((t_nodes == t_nodes[0]).all().item()) if t_nodes.numel() else True

    @staticmethod
    def _smooth_weight(pos_a, idx_a, pos_b, idx_b, max_norm):
        """get_edge_weight of the reference (all_atom_score_model.py:438-442, smooth_edges) for the edges (idx_a, idx_b);
        max_norm: a number or one value per edge.  Entries behind a list's device-side count hold no indices: clamped (their
        rows are never read)."""
        ia = idx_a.long().clamp(0, pos_a.shape[0] - 1)
        ib = idx_b.long().clamp(0, pos_b.shape[0] - 1)
        d = (pos_b[ib] - pos_a[ia]).norm(dim=-1)
        return 0.5 * (torch.cos(torch.clip(d * math.pi / max_norm, max=math.pi)) + 1.0)

    # ================================================================================================ prologue
    def _prologue(self, data, S, F, lig, dev, ll0, ll1):
        """Everything that depends on the times and the poses only, in ONE launch (ddp_step_prologue): t -> sigma (:244-245),
        the dynamic cross cutoff (:548-550), the graphs' sigma embedding (:371) and ligand centres (:571-576), the bond
        centres / directions of the torsion heads (:589-592,613-616,392,416), the bond rows of the ligand edge list (:462-468)."""
        m = self.m
        B = S.B
        a = L.PrologueArgs()
        ts = [data.complex_t[k] for k in ("tr", "rot", "tor", "sc_tor")]
        ts = [t if t.dtype == torch.float32 else t.float() for t in ts]
        keep = [ts]
        a.n_graphs = B
        for k, t in enumerate(ts):
            a.t[k], a.t_stride[k] = K._p(t), (t.stride(0) if t.numel() > 1 else 0)
        rng = None if m.confidence_mode else m._sigma_ranges()
        if m.confidence_mode:       # (:245) the times are used as they are
            F.sig = [t.expand(B).contiguous() for t in ts]
        elif rng is None:           # a foreign t_to_sigma: called as the reference calls it
            F.sig = [s.float().expand(B).contiguous() for s in m.t_to_sigma(*ts)]
        else:
            sig = torch.empty((4, B), device=dev)
            F.sig = [sig[k] for k in range(4)]
            for k in range(4):
                a.sig_min[k], a.sig_max[k] = rng[k]
        for k in range(4):
            a.sigma[k] = K._p(F.sig[k])
        F.cut = None
        if m.dynamic_max_cross:
            F.cut = torch.empty(B, device=dev)
            a.cut, a.cut_mul, a.cut_add = K._p(F.cut), 3.0, 20.0
        F.center = F.graph_emb = None
        if not m.confidence_mode:
            spec = m._sigma_spec(ts[0], dev)
            if spec[0] == "t":
                F.graph_emb = torch.empty((B, spec[4]), device=dev)
                a.graph_emb, a.sd, a.emb_scale, a.freq = K._p(F.graph_emb), spec[4], spec[2], K._p(spec[3])
            else:
                F.graph_emb = spec[1]
            keep.append(spec)
            F.center = torch.empty((B, 3), device=dev)
            a.lig_pos, a.graph_ptr, a.center = K._p(F.lpos), K._p(S.lay_l._ptr32), K._p(F.center)
        F.tor = F.sc = None
        for i, (name, st, pos) in enumerate((("tor", S.tor, F.lpos), ("sc", S.sc, F.apos))):
            if st is None:
                continue
            h = SimpleNamespace(st=st, bond_pos=torch.empty((st.T, 3), device=dev), bond_vec=torch.empty((st.T, 3), device=dev))
            b = a.bonds[i]
            b.pos, b.b0, b.b1, b.n = K._p(pos), K._p(st.flat32[:st.T]), K._p(st.flat32[st.T:]), st.T
            b.mid, b.vec = K._p(h.bond_pos), K._p(h.bond_vec)
            setattr(F, name, h)
        for i, (src, dst) in enumerate(((S.bond32[0], ll0), (S.bond32[1], ll1))):
            a.copy[i].src, a.copy[i].dst, a.copy[i].n = K._p(src), K._p(dst), S.E_bond
        L.check(L.load().ddp_step_prologue(C.byref(a), K.stream()), "ddp_step_prologue")
        F.keep_pro = (keep, a)

    # ================================================================================================ front
    def _front(self, data, S, lig, rec, atom, dev, mark, ff=None):
        m = self.m
        ns, B = m.ns, S.B
        F = SimpleNamespace()
        cnt = F.cnt = K.CountBlock(dev)
        lpos, rpos, apos = lig.pos.float().contiguous(), rec.pos.float().contiguous(), atom.pos.float().contiguous()
        F.lpos, F.rpos, F.apos = lpos, rpos, apos
        Nl, Nr, Na = S.Nl, S.Nr, S.Na
        lay_l, lay_r, lay_a = S.lay_l, S.lay_r, S.lay_a
        i32e = lambda n: torch.empty(n, dtype=torch.int32, device=dev)      # noqa: E731
        Eb = S.E_bond
        ll0, ll1 = i32e(S.cap_ll), i32e(S.cap_ll)     # ligand edges = bonds (first) + radius graph (:462-468)
        self._prologue(data, S, F, lig, dev, ll0, ll1)

        # node encoders, sigma embeddings and the per-node part of the edge-embedding MLPs' first Linear: one HIP launch
        # (ff: on a forked stream beside the neighbour searches; the edge embeddings below follow on that stream)
        nt = (lambda: m._node_tables(lig, rec, atom, dev))
        F.xl, F.xr, F.xa, pre = ff.run(0, nt) if ff is not None else nt()
        F.pre = pre
        mark("node_embed")
        sd_, dd, cd, nf = m.sigma_embed_dim, m.distance_embed_dim, m.cross_distance_embed_dim, m.in_lig_edge_features
        epk = {}
        for key, name, rbf0, rbf_n in (("ll", "lig_edge_embedding", nf + sd_, dd), ("rr", "rec_edge_embedding", sd_, dd),
                                       ("aa", "atom_edge_embedding", sd_, dd), ("lr", "lr_edge_embedding", sd_, cd),
                                       ("la", "la_edge_embedding", sd_, cd), ("ar", "ar_edge_embedding", sd_, dd)):
            epk[key] = m._edge_pack(name, slice(rbf0, rbf0 + rbf_n), dev)
        # bond-type columns of lig_edge_embedding's first Linear, [E_bond, ns]: fixed for a batch
        bond_attr = data["ligand", "ligand"].edge_attr
        bond_pre = m._cached("bond_pre", (bond_attr,), lambda: (bond_attr.float() @ epk["ll"].W1[:, :nf].t()).contiguous())

        # ---- the pose-dependent neighbour searches (:457,545-564,607,627): ONE batched count / scan / fill sequence, edge
        # counts stay on the device.  The atom kNN graph (:524) and everything derived from it is rebuilt only when atoms move
        ptr_l, ptr_r, ptr_a = lay_l._ptr32, lay_r._ptr32, lay_a._ptr32
        b32_l, b32_r, b32_a = G._batch32(lay_l, Nl), G._batch32(lay_r, Nr), G._batch32(lay_a, Na)
        rr = S.rr
        kk = m.atom_max_neighbors if m.atom_max_neighbors else 32
        aa = m._cached("aa", (atom.pos, atom.batch), lambda: G.knn_graph(apos, kk, lay_a))
        data["atom", "atom"].edge_index = aa
        F.aa = aa
        S.stats["E_aa"] = int(aa.shape[1])
        lr0, lr1, la0, la1 = i32e(S.cap_lr), i32e(S.cap_lr), i32e(S.cap_la), i32e(S.cap_la)
        jobs = [K.radius_job(lpos, ptr_l, lpos, b32_l, m.lig_max_radius, 33, _DROP_SELF, i32e(Nl), i32e(Nl + 1), base=Eb,
                             total=cnt["ll"], out_query=ll1, out_x=ll0, capacity=S.cap_ll)]
        cut = None
        if m.dynamic_max_cross:     # (:548-556) both point sets divided by 3 sigma_tr + 20, searched with r = 1
            cut = F.cut
            jobs.append(K.radius_job(rpos, ptr_r, lpos, b32_l, 1.0, 10000, 0, i32e(Nl), i32e(Nl + 1), total=cnt["lr"],
                                     out_query=lr0, out_x=lr1, capacity=S.cap_lr, graph_div=cut))
        else:
            jobs.append(K.radius_job(rpos, ptr_r, lpos, b32_l, m.cross_max_distance, 10000, 0, i32e(Nl), i32e(Nl + 1), total=cnt["lr"],
                                     out_query=lr0, out_x=lr1, capacity=S.cap_lr))
        jobs.append(K.radius_job(apos, ptr_a, lpos, b32_l, m.lig_max_radius, 10000, 0, i32e(Nl), i32e(Nl + 1), total=cnt["la"],
                                 out_query=la0, out_x=la1, capacity=S.cap_la, overflow=m.overflow_flag(dev)))
        # atoms with a ligand atom of their graph within the radius = the sources of ligand<-atom edges = the atoms an
        # atom<-ligand message reaches ("touched"): the same search with the roles swapped, count pass only
        F.touched = i32e(Na)
        jobs.append(K.radius_job(lpos, ptr_l, apos, b32_a, m.lig_max_radius, 10000, 0, F.touched))
        for name, st, pos, ptr_x in (("tor", S.tor, lpos, ptr_l), ("sc", S.sc, apos, ptr_a)):
            if st is None:
                continue
            # build_bond_conv_graph / build_sidechain_conv_graph (:586-636): atoms around the bond centres, default cap 32
            h = getattr(F, name)      # (bond centres and directions: ddp_step_prologue)
            h.q, h.x = i32e(st.cap), i32e(st.cap)
            jobs.append(K.radius_job(pos, ptr_x, h.bond_pos, G._batch32(st.lay, st.T), m.lig_max_radius, 32, 0, i32e(st.T),
                                     i32e(st.T + 1), total=cnt[name], out_query=h.q, out_x=h.x, capacity=st.cap))
            setattr(F, name, h)
        K.radius_search_jobs(jobs)
        F.keep = [jobs, cut]
        F.near_rows = i32e(Na)
        K.scan_jobs([K.scan_job(Na, flag=F.touched, lst=F.near_rows, total=cnt["near"])])
        mark("searches")

        # ---- which receptor-side work is the same in every graph (sampling batch: N poses of ONE complex at one time)
        shared0 = {}
        if m.share_layer0 and B > 1 and m.debug_conv_outputs is None and self._one_time(data, lig, rec, atom):
            ar = S.ar
            if S.num_flex > 0:      # side chains move per sample: only the receptor-receptor part can be shared
                sh_ = m._cached("shared0_rec", (rec.x, rec.pos, rr, atom.x, ar),
                                lambda: m._shared_receptor_side(B, rec, atom, rpos, apos, lay_r, lay_a, rr.long(), ar.long(), aa, atoms="static"))
            else:
                sh_ = m._cached("shared0", (rec.x, rec.pos, atom.x, atom.pos, rr, ar, aa),
                                lambda: m._shared_receptor_side(B, rec, atom, rpos, apos, lay_r, lay_a, rr.long(), ar.long(), aa))
            shared0 = {k: v for k, v in sh_.items() if v is not None}
        F.flex_static = shared0.pop("flex", None)
        F.shared0 = shared0

        def rows32(name, ei, k):   # int32 rows of a step-independent edge set (kept across calls), graph 0 only when shared
            r0, r1 = m._cached(name, (ei,), lambda: (_i32(ei[0]), _i32(ei[1])))
            n = shared0[k][1] if k in shared0 else ei.shape[1]
            return r0[:n], r1[:n]

        rr32, aa32, ar32 = rows32("rr32", rr, 6), rows32("aa32", aa, 3), rows32("ar32", S.ar, 5)

        # ---- edge embeddings + harmonics (the per-node `pre` tables came with the node encoders)
        F.e, F.sh = {}, {}
        calls = {   # one launch for the six edge sets (ddp_edge_featurize_jobs)
            "ll": ((epk["ll"], m.lig_distance_expansion, lpos, ll0, lpos, ll1, pre["ll"], ll0), dict(pre2=bond_pre, n_edges=S.cap_ll, cnt=cnt["ll"])),
            "rr": ((epk["rr"], m.rec_distance_expansion, rpos, rr32[0], rpos, rr32[1], pre["rr"], rr32[0]), {}),
            "aa": ((epk["aa"], m.lig_distance_expansion, apos, aa32[0], apos, aa32[1], pre["aa"], aa32[0]), {}),
            "lr": ((epk["lr"], m.cross_distance_expansion, lpos, lr0, rpos, lr1, pre["lr"], lr0), dict(n_edges=S.cap_lr, cnt=cnt["lr"])),
            "la": ((epk["la"], m.cross_distance_expansion, lpos, la0, apos, la1, pre["la"], la0), dict(n_edges=S.cap_la, cnt=cnt["la"])),
            "ar": ((epk["ar"], m.rec_distance_expansion, apos, ar32[0], rpos, ar32[1], pre["ar"], ar32[0]), {})}
        def feat():
            res = K.edge_featurize_jobs(list(calls.values()))
            if m.smooth_edges:
                # get_edge_weight (:438-442): the fc output of an edge times 0.5 (cos(min(d pi / max_norm, pi)) + 1).  A message is
                # linear in the fc output and in the edge's harmonics alike: the weight goes into the harmonics (plain PyTorch
                # launches - a rarely used option).  atom-receptor edges: weight 1 (:580)
                cut_e = (lambda: F.cut[b32_l.long()[lr0.long().clamp(0, Nl - 1)]]) if m.dynamic_max_cross else (lambda: m.cross_max_distance)
                for key, mx in (("ll", m.lig_max_radius), ("rr", m.rec_max_radius), ("aa", m.lig_max_radius), ("lr", None), ("la", m.lig_max_radius)):
                    (_, _, pa, ia, pb, ib, _, _), _ = calls[key]
                    sh_k = res[list(calls).index(key)][1]
                    if sh_k.shape[0] > 0:
                        sh_k.mul_(self._smooth_weight(pa, ia[:sh_k.shape[0]], pb, ib[:sh_k.shape[0]], cut_e() if mx is None else mx).unsqueeze(1))
            return res
        if ff is not None:
            ff.resync(0)      # the node tables' stream also waits for the searches; the views below are built beside it
        for key, (o_, s_) in zip(calls, ff.run(0, feat) if ff is not None else feat()):
            F.e[key], F.sh[key] = o_, s_
        mark("edge_featurize")

        # ---- CSR views per conv direction (receiver = edge_index[0] of the conv call) and source-ordered views of the
        # factorised convs: two batched grouping calls for everything pose-dependent, the rest is static
        def static_csr(name, k, ei, recv, src, n):
            """CSR view of a step-independent edge set, kept across calls; edge ids modulo the per-graph edge count when
            the edge embeddings exist for graph 0 only."""
            if k in shared0:
                e0 = shared0[k][1]

                def modded():
                    c = G.build_csr(recv, src, n)
                    return G.CSR(c.n_edges, c.recv, c.src, (c.eid % e0).contiguous(), c.rowptr)
                return m._cached(f"{name}_mod{e0}", (ei,), modded)
            return m._cached(name, (ei,), lambda: G.build_csr(recv, src, n))

        c = {}
        g1 = []
        aa_per_step = S.num_flex > 0 and aa.shape[1] > 0     # the atom kNN graph is rebuilt every step: its views join the batched calls
        if aa_per_step:
            v = SimpleNamespace(rp=i32e(Na + 1), perm=i32e(aa.shape[1]), key=i32e(aa.shape[1]), o0=i32e(aa.shape[1]))
            g1.append(K.group_job(aa32[0], aa.shape[1], Na, [aa32[1]], v.rp, v.perm, v.key, [v.o0], i32e(Na + aa.shape[1])))
            c[3] = G.CSR(int(aa.shape[1]), v.key, v.o0, v.perm, v.rp)
        else:
            c[3] = static_csr("c_aa", 3, aa, aa[0], aa[1], Na)
        c[5] = static_csr("c_ar", 5, S.ar, S.ar[0], S.ar[1], Na)
        c[6] = static_csr("c_rr", 6, rr, rr[0], rr[1], Nr)
        c[8] = static_csr("c_ra", 8, S.ar, S.ar[1], S.ar[0], Nr)
        v = SimpleNamespace(rp=i32e(Nl + 1), perm=i32e(S.cap_ll), key=i32e(S.cap_ll), o0=i32e(S.cap_ll))
        g1.append(K.group_job(ll0, S.cap_ll, Nl, [ll1], v.rp, v.perm, v.key, [v.o0], i32e(Nl + S.cap_ll), n_dev=cnt["ll"]))
        c[0] = EdgeView(S.cap_ll, v.key, v.o0, v.perm, rowptr=v.rp, cnt=cnt["ll"])
        for k, q0, x1, cap, name in ((1, lr0, lr1, S.cap_lr, "lr"), (2, la0, la1, S.cap_la, "la")):
            rp = i32e(Nl + 1)       # (query-major search output: already grouped by the receiving ligand atom)
            g1.append(K.group_job(q0, cap, Nl, [], rp, scratch=i32e(Nl), n_dev=cnt[name]))
            c[k] = EdgeView(cap, q0, x1, G.iota32(cap, dev), rowptr=rp, cnt=cnt[name])
        for k, key, pay, cap, n_keys, name in ((4, la1, la0, S.cap_la, Na, "la"), (7, lr1, lr0, S.cap_lr, Nr, "lr")):
            v = SimpleNamespace(rp=i32e(n_keys + 1), perm=i32e(cap), key=i32e(cap), o0=i32e(cap))
            g1.append(K.group_job(key, cap, n_keys, [pay], v.rp, v.perm, v.key, [v.o0], i32e(n_keys + cap), n_dev=cnt[name]))
            c[k] = EdgeView(cap, v.key, v.o0, v.perm, rowptr=v.rp, cnt=cnt[name])
        for h, name in ((F.tor, "tor"), (F.sc, "sc")):
            if h is not None:
                h.rp = i32e(h.st.T + 1)
                g1.append(K.group_job(h.q, h.st.cap, h.st.T, [], h.rp, scratch=i32e(h.st.T), n_dev=cnt[name]))
                h.csr = EdgeView(h.st.cap, h.q, h.x, G.iota32(h.st.cap, dev), rowptr=h.rp, cnt=cnt[name])
        K.group_jobs(g1)
        F.keep.append(g1)
        F.c = c
        n_src = {"l": Nl, "a": Na, "r": Nr}
        so = {}
        F.fact = set()
        if m.factorize_min_degree > 0:
            # Source-node factorised convs (packing.faster_tp_spec(factorized=True)): every conv whose edge set has several
            # edges per source node - all but receptor<-atom (one edge per atom).  Their edges are listed in source order.
            F.fact = {0, 1, 2, 3, 4, 5, 6, 7}
            for k in (3, 5, 6):
                if k == 3 and aa_per_step:
                    continue
                so[k] = m._cached(f"so_{k}", (c[k].src, c[k].eid), lambda k=k: G.source_order(c[k], n_src[SRC_TYPE[k]]))
            # ligand<-receptor / ligand<-atom in source order = the receptor<-ligand / atom<-ligand CSR views read the other way
            so[1] = EdgeView(S.cap_lr, c[7].src, c[7].recv, c[7].eid, pos=c[7].eid, cnt=cnt["lr"])
            so[2] = EdgeView(S.cap_la, c[4].src, c[4].recv, c[4].eid, pos=c[4].eid, cnt=cnt["la"])
            g2 = []
            if aa_per_step:
                E3 = c[3].n_edges
                v = SimpleNamespace(rp=i32e(Na + 1), perm=i32e(E3), key=i32e(E3), o0=i32e(E3), o1=i32e(E3))
                g2.append(K.group_job(c[3].src, E3, Na, [c[3].recv, c[3].eid], v.rp, v.perm, v.key, [v.o0, v.o1], i32e(Na + E3)))
                so[3] = G.SourceOrder(E3, v.o0, v.key, v.o1, v.perm)
            for k, cap, name in ((0, S.cap_ll, "ll"), (4, S.cap_la, "la"), (7, S.cap_lr, "lr")):
                v = SimpleNamespace(rp=i32e(Nl + 1), perm=i32e(cap), key=i32e(cap), o0=i32e(cap), o1=i32e(cap))
                g2.append(K.group_job(c[k].src, cap, Nl, [c[k].recv, c[k].eid], v.rp, v.perm, v.key, [v.o0, v.o1], i32e(Nl + cap),
                                      n_dev=cnt[name]))
                so[k] = EdgeView(cap, v.o0, v.key, v.o1, pos=v.perm, cnt=cnt[name])
            K.group_jobs(g2)
            F.keep.append(g2)
        F.so = so
        if ff is not None:
            ff.join(only=0)
        mark("views")
        return F

    # ================================================================================================ index lists
    def _lists(self, S, F, dev, side=None):
        """Index lists of the exact work eliminations, built on the device from graph STRUCTURE only (no features)."""
        m = self.m
        L_, B = m.num_conv_layers, S.B
        Nl, Nr, Na = S.Nl, S.Nr, S.Na
        cnt, c, so = F.cnt, F.c, F.so
        i32e = lambda n: torch.empty(n, dtype=torch.int32, device=dev)      # noqa: E731
        # (every zero-initialised mask of this function is a slice of ONE zero-filled block: one launch)
        zpool = [torch.zeros(6 * (Na + Nr) + Na + 64, dtype=torch.int32, device=dev), 0]
        F.keep.append(zpool)      # (allocated on this stream, read and written by the forked part: alive until the forward has joined it)

        def i32z(n):
            if zpool[1] + n > zpool[0].numel():
                return torch.zeros(n, dtype=torch.int32, device=dev)
            v = zpool[0][zpool[1]:zpool[1] + n]
            zpool[1] += (n + 3) & ~3
            return v
        F.pruned, F.pruned_so, F.rows_a = {}, {}, {}
        F.clean1 = None
        n_of = {"l": Nl, "a": Na, "r": Nr}
        dbg = m.debug_conv_outputs is not None
        E_aa = c[3].n_edges

        def prune():
            # ---- Dead-output elimination over the last layers.  What is read after the last layer: all ligand features (heads),
            # with flexible side chains the atom features around the flexible bonds (side-chain torsion head), nothing of the
            # receptor.  Walking backwards, a layer's receptor-side convs only have to produce the rows that are still read
            # (by the residual of a needed node or as source / receiver of a kept edge of the next layer), so their edge lists
            # are restricted to the edges that END in a needed node - exact, the other rows of x are simply left stale.
            # Without flexible side chains this prunes layer L-2 (its atom outputs feed only the final ligand<-atom conv), with
            # them layers L-1 and L-2; earlier layers feed (almost) everything and run in full.  Layer 0 is never touched.
            prune_on = (m.prune_last_receptor_layer and L_ >= 2 and not m.confidence_mode and not dbg and E_aa >= m.plan_min_edges
                        and not (m.flexible_sidechains and F.sc is None))
            if prune_on:
                layers = [l for l in (L_ - 1, L_ - 2) if l >= 1 and (l != L_ - 1 or m.flexible_sidechains)]
                need = {"a": i32z(Na), "r": i32z(Nr)}
                if F.sc is not None:    # the side-chain head reads the atoms of the flexible bonds and the atoms around them
                    K.mark_jobs([K.mark_job(need["a"], F.sc.st.flat32, 2 * F.sc.st.T),
                                 K.mark_job(need["a"], F.sc.csr.src, F.sc.csr.n_edges, cnt["sc"])])
                else:                   # layer L-1 (ligand-receiving convs only): sources of ligand<-atom / ligand<-receptor
                    K.mark_jobs([K.mark_job(need["a"], c[2].src, c[2].n_edges, cnt["la"]),
                                 K.mark_job(need["r"], c[1].src, c[1].n_edges, cnt["lr"])])
                for l in layers:
                    act = {"a": m.flexible_sidechains or l != L_ - 1}
                    act["r"] = act["a"] and l != L_ - 1
                    scans, copies, pl = [], [], {}
                    for rt in ("a", "r"):
                        if not act[rt]:
                            continue
                        for k in RECV_OF[rt]:
                            full = c[k]
                            if full.n_edges == 0:
                                continue
                            name = f"p{l}_{k}"
                            n = n_of[rt]
                            v = EdgeView(full.n_edges, i32e(full.n_edges), i32e(full.n_edges), i32e(full.n_edges), rowptr=i32e(n + 1), cnt=cnt[name])
                            scans.append(K.scan_job(n, flag=need[rt], rowptr=full.rowptr, excl=v.rowptr, total=cnt[name]))
                            copies.append(K.rowcopy_job(n, need[rt], full.rowptr, v.rowptr, [full.recv, full.src, full.eid], [v.recv, v.src, v.eid]))
                            pl[k] = v
                    K.scan_jobs(scans)
                    K.rowcopy_jobs(copies)
                    F.pruned[l] = pl
                    # source rows of this layer's factorised convs per source-node array (stage A runs on them only) ...
                    rows_mask = {"a": i32z(Na), "r": i32z(Nr)}
                    marks = []
                    for k in range(9):
                        rt, st_ = RECV_TYPE[k], SRC_TYPE[k]
                        if st_ == "l" or k == 2 or k not in F.fact or (rt != "l" and not act[rt]):
                            continue
                        vw = pl.get(k, c[k])
                        if vw.n_edges > 0:
                            marks.append(K.mark_job(rows_mask[st_], vw.src, vw.n_edges, vw.cnt))
                    # ... and the rows the NEXT (earlier) pruned layer has to produce: the needed rows themselves (residual) and the
                    # sources of every kept edge of this layer
                    nxt = None
                    if l != layers[-1]:
                        nxt = {"a": need["a"].clone(), "r": need["r"].clone()}
                        for k in range(9):
                            rt, st_ = RECV_TYPE[k], SRC_TYPE[k]
                            if st_ == "l" or (rt != "l" and not act[rt]):
                                continue
                            vw = pl.get(k, c[k])
                            if vw.n_edges > 0:
                                marks.append(K.mark_job(nxt[st_], vw.src, vw.n_edges, vw.cnt))
                    K.mark_jobs(marks)
                    lists = {}
                    scans = []
                    for t in ("a", "r"):
                        rows = i32e(n_of[t])
                        scans.append(K.scan_job(n_of[t], flag=rows_mask[t], lst=rows, total=cnt[f"rows{l}_{t}"]))
                        lists[t] = (rows, cnt[f"rows{l}_{t}"])
                    K.scan_jobs(scans)
                    F.rows_a[l] = lists
                    if nxt is not None:
                        need = nxt
                # source-ordered views of the pruned factorised convs: one batched grouping call
                gj = []
                for l, pl in F.pruned.items():
                    for k, v in pl.items():
                        if k in F.fact:
                            n_keys = n_of[SRC_TYPE[k]]
                            w = SimpleNamespace(rp=i32e(n_keys + 1), perm=i32e(v.n_edges), key=i32e(v.n_edges), o0=i32e(v.n_edges), o1=i32e(v.n_edges))
                            gj.append(K.group_job(v.src, v.n_edges, n_keys, [v.recv, v.eid], w.rp, w.perm, w.key, [w.o0, w.o1],
                                                  i32e(n_keys + v.n_edges), n_dev=v.cnt))
                            F.pruned_so.setdefault(l, {})[k] = EdgeView(v.n_edges, w.o0, w.key, w.o1, pos=w.perm, cnt=v.cnt)
                K.group_jobs(gj)
                F.keep.append(gj)

        def flex0():
            # ---- Layer 0 with flexible side chains (N poses of one complex, only the flexible residues' side chains differ between
            # the samples).  The node features entering layer 0 are the same in every sample (one diffusion time), so the message
            # of an edge is the same wherever its two ends sit where they sit in sample 0.  The atom graph is a kNN graph: the edge
            # r <- q exists because r is one of the k nearest atoms of the QUERY q (reference :524; receivers have no fixed degree).
            # "Moved" = position differs from sample 0's.  (1) If q and every member of q's list in sample s AND in sample 0 is
            # unmoved in s, both lists are the k nearest unmoved atoms around q at identical distances: equal, entry by entry.
            # (2) A receiver r none of whose incoming edges - in sample s or in sample 0 - leaves a query failing (1) therefore has
            # the same incoming edges, in the same order (edges are listed by query), with the same geometry as its copy in sample 0
            # (a moved r fails: every query that found it fails (1)).  Such receivers read sample 0's messages through a row map;
            # sample 0 and the marked receivers of the other samples are computed (pruned-list machinery: stage A runs on the kept
            # edges' source rows only).  atom<-receptor (one edge per atom) uses the same receiver marks (a superset of the moved
            # atoms); receptor<-atom: a residue needs its own messages if one of its atoms moved.
            # Bitwise the general path (tests/test_gpu_parity.py::test_flexible_layer0_sharing_is_exact).
            F.flex0 = None
            fx = getattr(F, "flex_static", None)
            kk = m.atom_max_neighbors if m.atom_max_neighbors else 32
            if (fx is not None and m.share_flex_layer0 and S.num_flex > 0 and not dbg and 0 not in F.pruned and L_ >= 2
                    and 3 in F.fact and 5 in F.fact and E_aa == Na * kk and E_aa % B == 0 and E_aa >= m.plan_min_edges
                    and E_aa * m.ns >= m.flex_share_min_work):
                na, e_ar, nr = fx
                e0 = E_aa // B
                qdirty, dirty, need_r = i32z(Na), i32z(Na), i32z(Nr)
                # pass 1, per QUERY atom q (the source end: edge r <- q exists because r is one of q's k nearest): q's list may differ
                # from sample 0's, or carries other geometry, if q or a member of its list - here or in sample 0 - moved
                K.flex_mark(c[3].src, c[3].recv, E_aa, e0, na, na, qdirty, pos=F.apos, a_too=True, ref_list=True)
                # pass 2, per RECEIVER r: its incoming edges are those of its copy in sample 0 unless one of them - here or there -
                # leaves such a query (r moved itself: then every query that found it is marked, and r with them)
                K.flex_mark(c[3].recv, c[3].src, E_aa, e0, na, na, dirty, flag=qdirty, a_too=True, ref_list=True)
                K.flex_mark(c[8].recv, c[8].src, c[8].n_edges, e_ar, nr, na, need_r, pos=F.apos)
                scans, copies, pl, rowmaps = [], [], {}, {}
                for k, need, n in ((3, dirty, Na), (5, dirty, Na), (8, need_r, Nr)):
                    full = c[k]
                    v = EdgeView(full.n_edges, i32e(full.n_edges), i32e(full.n_edges), i32e(full.n_edges), rowptr=i32e(n + 1), cnt=cnt[f"fx_{k}"])
                    scans.append(K.scan_job(n, flag=need, rowptr=full.rowptr, excl=v.rowptr, total=cnt[f"fx_{k}"]))
                    copies.append(K.rowcopy_job(n, need, full.rowptr, v.rowptr, [full.recv, full.src, full.eid], [v.recv, v.src, v.eid]))
                    pl[k] = v
                K.scan_jobs(scans)
                K.rowcopy_jobs(copies)
                for k, need, n_per in ((3, dirty, na), (5, dirty, na), (8, need_r, nr)):
                    rowmaps[k] = i32e(c[k].n_edges)
                    K.fallback_rowmap(need, c[k].recv, c[k].rowptr, pl[k].rowptr, c[k].n_edges, n_per, rowmaps[k])
                # source rows stage A has to produce: atoms for atom<-atom, residues for atom<-receptor and ligand<-receptor
                rows_mask = {"a": i32z(Na), "r": i32z(Nr)}
                marks = [K.mark_job(rows_mask["a"], pl[3].src, pl[3].n_edges, pl[3].cnt),
                         K.mark_job(rows_mask["r"], pl[5].src, pl[5].n_edges, pl[5].cnt)]
                if 1 in F.fact and c[1].n_edges > 0:
                    marks.append(K.mark_job(rows_mask["r"], c[1].src, c[1].n_edges, c[1].cnt))
                K.mark_jobs(marks)
                lists, scans = {}, []
                for t in ("a", "r"):
                    rows = i32e(n_of[t])
                    scans.append(K.scan_job(n_of[t], flag=rows_mask[t], lst=rows, total=cnt[f"rows0_{t}"]))
                    lists[t] = (rows, cnt[f"rows0_{t}"])
                K.scan_jobs(scans)
                gj, pso = [], {}
                for k in (3, 5):
                    v = pl[k]
                    n_keys = n_of[SRC_TYPE[k]]
                    w = SimpleNamespace(rp=i32e(n_keys + 1), perm=i32e(v.n_edges), key=i32e(v.n_edges), o0=i32e(v.n_edges), o1=i32e(v.n_edges))
                    gj.append(K.group_job(v.src, v.n_edges, n_keys, [v.recv, v.eid], w.rp, w.perm, w.key, [w.o0, w.o1],
                                          i32e(n_keys + v.n_edges), n_dev=v.cnt))
                    pso[k] = EdgeView(v.n_edges, w.o0, w.key, w.o1, pos=w.perm, cnt=v.cnt)
                K.group_jobs(gj)
                F.keep.append(gj)
                F.pruned[0], F.pruned_so[0], F.rows_a[0] = pl, pso, lists
                F.flex0 = SimpleNamespace(dirty=dirty, need_r=need_r, rowmaps=rowmaps, na=na, e0=e0)

        def clean1():
            # ---- Layer 1, atom<-atom, sampling batches of one rigid complex (shared0 has conv 3): after the shared layer 0 an
            # atom's features differ between the samples only if an atom<-ligand message reached it ("touched", the atoms within
            # 5 A of that sample's ligand, ~15 %).  A layer-1 atom<-atom message between two untouched atoms is therefore the same
            # in every sample: those messages are computed ONCE on the complex's own edge list (e0 edges, rows [E, E + e0) of
            # the message array) and the segmented mean reads them through a row map; only the edges with a touched end are
            # computed per sample (source-ordered sub-list, stage A on their source rows only).  Messages of a clean pair are
            # bitwise those the general path computes (same inputs, per-edge arithmetic), the mean sums the same values in the
            # same order: the result is bitwise the general path's (GPU test).  Layer 1 must not be one of the pruned layers.
            # With flexible side chains (F.flex0): the same, with "touched" widened by the receivers layer 0 computed per sample.
            if (m.share_clean_layer1 and (3 in F.shared0 or F.flex0 is not None) and 3 in so and L_ >= 4 and E_aa > 0
                    and E_aa >= m.plan_min_edges and not dbg and 1 not in F.pruned):
                if F.flex0 is not None:
                    n0, e0 = F.flex0.na, F.flex0.e0
                    F.touched_l1 = F.touched + F.flex0.dirty
                else:
                    n0, e0, _ = F.shared0[3]
                    F.touched_l1 = F.touched
                so3 = so[3]
                d = SimpleNamespace(recv=i32e(E_aa), src=i32e(E_aa), eid=i32e(E_aa), pos=i32e(E_aa))
                K.select_jobs([K.select_job(E_aa, F.touched_l1, so3.recv, F.touched_l1, so3.src, [so3.recv, so3.src, so3.eid, so3.pos],
                                            [d.recv, d.src, d.eid, d.pos], cnt["dirty"], i32e(2 * ((E_aa + 2047) // 2048) + 1))])
                so_d = EdgeView(E_aa, d.recv, d.src, d.eid, pos=d.pos, cnt=cnt["dirty"])
                mask, rows_d = i32z(Na), i32e(Na)
                K.mark_jobs([K.mark_job(mask, d.src, E_aa, cnt["dirty"])])
                K.scan_jobs([K.scan_job(Na, flag=mask, lst=rows_d, total=cnt["rows_dirty"])])
                rowmap, rows_v = i32e(E_aa), i32e(n0)
                K.clean_pair_maps(F.touched_l1, c[3].recv, c[3].src, E_aa, e0, B, n0, rowmap, rows_v, rowptr=c[3].rowptr if F.flex0 is not None else None)
                so_v = m._cached(f"so_v{e0}", (so3.pos,), lambda: G.SourceOrder(e0, so3.recv[:e0], so3.src[:e0], so3.eid[:e0], (so3.pos[:e0] + E_aa).contiguous()))
                F.clean1 = SimpleNamespace(so_d=so_d, rows_d=rows_d, rows_d_cnt=cnt["rows_dirty"], rowmap=rowmap, so_v=so_v, rows_v=rows_v,
                                           E=E_aa, e0=e0, n0=n0)

        # The three parts read graph structure only.  Rigid receptor: all of them on the forked stream (`side`, _forward); with
        # flexible side chains layer 0 reads the flex0 lists: those and the layer-1 lists (which build on them) stay on the
        # current stream, the pruned lists of the last layers - independent of both - go to the forked one
        if side is not None and S.num_flex > 0 and m.fork_lists_flex:
            side(prune)
            flex0()
            clean1()
        elif side is not None and S.num_flex == 0:
            side(lambda: (prune(), flex0(), clean1()))
        else:
            prune()
            flex0()
            clean1()

    def _exact(self, F):
        """Test mode (`model.exact_sizes`): every device-side count is read back and every list cut to its actual length, so
        that all kernels run with host-known sizes and exact grids - the results must not depend on the capacities."""
        vals = F.cnt.values()
        by_ptr = {F.cnt.block[i:i + 1].data_ptr(): vals[k] for k, i in F.cnt.index.items()}

        def cut(v):
            if v is None or v.cnt is None:
                return v
            n = by_ptr[v.cnt.data_ptr()]
            return EdgeView(n, v.recv[:n], v.src[:n], v.eid[:n], v.rowptr, v.pos[:n] if v.pos is not None else None, None)

        F.c = {k: cut(v) for k, v in F.c.items()}
        F.so = {k: cut(v) for k, v in F.so.items()}
        F.pruned = {l: {k: cut(v) for k, v in pl.items()} for l, pl in F.pruned.items()}
        F.pruned_so = {l: {k: cut(v) for k, v in pl.items()} for l, pl in F.pruned_so.items()}
        for h in (F.tor, F.sc):
            if h is not None:
                h.csr = cut(h.csr)
        if F.clean1 is not None:
            F.clean1.so_d = cut(F.clean1.so_d)
        F.exact = by_ptr

    # ================================================================================================ layers
    def _stage_a(self, l, convs, x, n_rows, rows=None, rows_cnt=None, dense_rows=None):
        """Stage A of the factorised convs of layer `l` that read the same source rows: ONE ddp_stage_a launch
        rows[(conv, slot)] = x[:, scalars(slot)] @ Wg[(conv, slot)] = [G | Gb | pad] for all of them (weight-stationary
        fp32-MFMA kernel, csrc/ddp_gemm.hip; bound by the HBM write of G).  convs: [(k, TensorProductConvLayer)].  With a row
        list only the listed rows of the [n_rows]-row G arrays are computed.  Returns {(k, slot): G rows}."""
        m = self.m
        # (G leaves stage A in the layout the conv kernel of this layer reads: plane form for ddp_conv_rows, launch.rows_mode)
        rows_k = all(K.rows_mode(conv.packed_g(x.device)) for _, conv in convs)
        key = (l, tuple(k for k, _ in convs), rows_k)
        ent = m._stage_a_stacks.get(key)
        if ent is None or ent[0].device != x.device:
            Ws, meta, ghs, lds, fmts = [], [], [], [], set()
            for k, conv in convs:
                pk = conv.packed_g(x.device)
                for slot in (0, 1):
                    if pk.wg[slot] is not None:
                        Ws.append(pk.wgh[slot] if rows_k else pk.wg[slot])
                        meta.append((k, slot, pk.g_in_off[slot]))
                        if rows_k:
                            ghs.append(pk.gh_groups[slot])
                            lds.append(pk.gh_ld[slot])
                            fmts.add(pk.gh_fmt)
            assert len(fmts) <= 1 and len(set(lds)) <= 1, "the convs of one stage-A launch write G rows of one plane form and length"
            gfmt = fmts.pop() if fmts else 0
            Wst = torch.stack(Ws).contiguous()
            # (the bf16x3 split of the weights - 1.5 x their size and three copy kernels - only when that option is on)
            ent = (Wst, meta, (C.c_int32 * len(meta))(*[mm[2] for mm in meta]), P.split_bf16x3(Wst) if (m.stage_a_bf16x3 and not rows_k) else None,
                   (P.split_h2(Wst, unified_scale=P.GH_SW) if rows_k else P.split_h2(Wst)) if (m.stage_a_h2 or rows_k) else None,
                   torch.stack([P.gh_dest_table(ws, (convs[0][1].spec_g.hid + 7) // 8, Wst.shape[2], fmt=gfmt) for ws in ghs]).contiguous().to(x.device) if rows_k else None,
                   (gfmt, lds[0] if lds else Wst.shape[2]))
            m._stage_a_stacks[key] = ent
        if m.stage_a_bf16x3 and ent[3] is None and not rows_k:
            ent = m._stage_a_stacks[key] = ent[:3] + (P.split_bf16x3(ent[0]),) + ent[4:]
        if m.stage_a_h2 and ent[4] is None:
            ent = m._stage_a_stacks[key] = ent[:4] + (P.split_h2(ent[0], unified_scale=P.GH_SW) if rows_k else P.split_h2(ent[0]),) + ent[5:]
        gh, (gfmt, g_ld) = ent[5], ent[6]
        ent = ent[:5]
        Wst, meta, offs, W3, Wh = ent
        nb = len(meta)
        if nb > L.DDP_MAX_GEMM_BATCH:
            raise L.DdpError("more (conv, slot) pairs per source array than DDP_MAX_GEMM_BATCH")
        Gall = torch.empty((nb, n_rows, g_ld), device=x.device, dtype=torch.float32)   # 128-byte aligned rows (plane form 1: 3 / 4 of the columns)
        prof = K.profiler(hbm=True)
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        n_list = n_rows if rows is None else int(rows.shape[0])
        if gh is not None:
            K.stage_a(x, n_list, offs, nb, Wst, Gall, rows=rows, rows_cnt=rows_cnt, out_rows=n_rows, Wh=Wh, gh=gh, gh_fmt=gfmt, ldo=g_ld)
        else:
            K.stage_a(x, n_list, offs, nb, Wst, Gall, rows=rows, rows_cnt=rows_cnt, out_rows=n_rows, W3=W3 if m.stage_a_bf16x3 else None,
                      Wh=Wh if (m.stage_a_h2 and K.CONV_H2 and not m.stage_a_bf16x3) else None)
        if prof is not None:
            e1.record()
            # algorithmic bytes: the G rows written once + the scalar columns of x read once per product + the weights
            row_bytes = 4.0 * nb * (g_ld + Wst.shape[1])

            def nbytes(n_list=n_list, rows_cnt=rows_cnt, row_bytes=row_bytes, wn=Wst.numel()):
                n = n_list if rows_cnt is None else min(n_list, int(rows_cnt.item()))
                return row_bytes * n + 4.0 * wn
            use_h2 = gh is not None or (m.stage_a_h2 and K.CONV_H2 and not m.stage_a_bf16x3)
            prof.hbm.setdefault("ddp_stage_a_h2_kernel" if use_h2 else "ddp_stage_a_mfma_kernel", []).append((e0, e1, nbytes))
        return {(k, slot): Gall[i] for i, (k, slot, _) in enumerate(meta)}

    def _layers(self, S, F, dev, mark):
        """The conv layers (reference :271-324).  Per layer: stage A of the factorised convs, the 32-edge conv launch (all
        factorised convs), the direct conv launch (receptor<-atom), the segmented means in the reference's summation order.

        Three launch orders over the same kernels with the same arguments (same bits):
          serial     stage A(l) -> conv32(l) -> direct(l) -> means(l)                      (debug hooks, section timer, resident groups)
          forked     small batches: direct(l) | stage-A groups side by side, then conv32(l) (see _Fork)
          pipelined  large batches: the direct conv of layer l + 1 (receptor<-atom, one 130-KB workgroup per CU) needs x(l + 1) only -
                     neither stage A nor the 32-edge launch - so it starts on a stream of its own as soon as mean{atom}(l) and
                     mean{rec}(l) are queued and is joined before the means of layer l + 1 update x in place; the window between two
                     32-edge launches is [mean{rec} -> stage A{rec rows}] | [mean{lig} -> stage A{lig rows}] | [mean{atom} -> stage
                     A{atom rows}] as parallel branches of the captured step.  (Round 4 measured five other overlapped orders
                     against this one - DESIGN.md section 4.8 keeps the numbers; their code is gone.)"""
        m = self.m
        ns, L_, B, ldx = m.ns, m.num_conv_layers, S.B, m._ldx
        Nl, Nr, Na = S.Nl, S.Nr, S.Na
        c, so = F.c, F.so
        xl, xr, xa = F.xl, F.xr, F.xa
        e, sh = F.e, F.sh
        # conv k of a layer: (receiver x, source x, edge embedding, harmonics)
        arr = {0: (xl, xl, "ll"), 1: (xl, xr, "lr"), 2: (xl, xa, "la"), 3: (xa, xa, "aa"), 4: (xa, xl, "la"), 5: (xa, xr, "ar"),
               6: (xr, xr, "rr"), 7: (xr, xl, "lr"), 8: (xr, xa, "ar")}
        nodes = {"l": (xl, Nl), "a": (xa, Na), "r": (xr, Nr)}
        dbg = m.debug_conv_outputs
        exact = getattr(F, "exact", None)
        prof_on = K.profiler() is not None
        # small batches: independent launches of a layer side by side (see _Fork); decided by the batch's size, not its content
        plain = m.before_layers is not None
        fork = self._fork(dev) if (m.concurrent_small_batches and Na <= m.concurrent_max_atoms and not plain) else None
        # (the pipelined order relies on the direct convs feeding the RECEPTOR mean only: every other conv must be factorised)
        pipelined = (fork is None and m.overlap_direct_conv and not plain and dbg is None and m.section_timer is None and L_ >= 2
                     and F.fact == {0, 1, 2, 3, 4, 5, 6, 7})
        side = self._fork(dev) if pipelined else None
        lists_fork = getattr(F, "lists_fork", None)     # the index lists still running on a stream of their own (_forward)

        def join_lists():
            nonlocal lists_fork
            if lists_fork is not None:
                lists_fork.join(only=4)
                lists_fork = None
        if not pipelined and fork is None:
            join_lists()      # (the serial order: before anything else)

        def plan(l):
            """Host side of layer l: per conv its CSR view (mean / direct conv), its source-ordered view and the rows stage A has
            to produce, grouped by (source-node array, row set)."""
            P = SimpleNamespace(l=l, spec=m._layer_specs[l], spec_g=m._layer_specs_g[l], gmap={}, keep=[], msgs={})
            do_atom = m.flexible_sidechains or l != L_ - 1
            P.active = active = {"l": True, "a": do_atom, "r": do_atom and l != L_ - 1}
            P.shared = shared = F.shared0 if l == 0 else {}
            pl, pl_so = F.pruned.get(l, {}), F.pruned_so.get(l, {})
            P.c1 = c1 = F.clean1 if (l == 1 and F.clean1 is not None and active["a"]) else None
            P.per, P.groups = per, groups = {}, {}
            for k in range(9):
                rt, st_ = RECV_TYPE[k], SRC_TYPE[k]
                if not active[rt]:
                    continue
                csr, so_k = c[k], so.get(k) if k in F.fact else None
                x_src = arr[k][1]
                rows = ("all", None, None, nodes[st_][1])                 # (group id, row list, its device count, rows of G)
                if k in pl:             # edges that end in a node the final layers read
                    csr, so_k = pl[k], pl_so.get(k)
                    if st_ != "l":
                        r_, rc = F.rows_a[l][st_]
                        rows = (f"pr{l}", r_, rc, nodes[st_][1])
                elif k in shared:       # graph 0's edges = a prefix of both orderings, graph 0's nodes = a prefix of x
                    n0, e0, ns0 = shared[k]
                    csr = csr.prefix(e0, n0)
                    so_k = so_k.prefix(e0) if so_k is not None else None
                    x_src = x_src[:ns0]
                    rows = ("g0", None, None, ns0)
                elif k == 2 and so_k is not None:    # ligand<-atom: only the atoms near a ligand occur as sources
                    rows = ("near", F.near_rows, F.cnt["near"], Na)
                    if exact is not None:
                        rows = ("near", F.near_rows[:exact[F.cnt["near"].data_ptr()]], None, Na)
                elif l in F.pruned and st_ != "l" and so_k is not None:   # an unpruned conv of a pruned layer (ligand<-receptor)
                    r_, rc = F.rows_a[l][st_]
                    rows = (f"pr{l}", r_, rc, nodes[st_][1])
                if exact is not None and rows[2] is not None:
                    rows = (rows[0], rows[1][:exact[rows[2].data_ptr()]], None, rows[3])
                per[k] = (csr, so_k, x_src)
                if so_k is not None and csr.n_edges > 0 and not (k == 3 and c1 is not None):
                    groups.setdefault((st_, rows[0]), (x_src, rows, []))[2].append((k, m.conv_layers[9 * l + k]))
            return P

        def node_bytes(l, k):     # algorithmic node bytes of a conv call (profiler only)
            d_in = P_.irreps_dim(P_.irreps_muls(ns, m.nv, l))
            return 4.0 * (nodes[SRC_TYPE[k]][1] * d_in + nodes[RECV_TYPE[k]][1] * m._layer_specs[l].d_out)

        def stage_a(P, which, forked=None):
            """Stage A of the groups whose source-node type is in `which`: one batched product per (source-node array, row set);
            then, for atom sources at layer 1, the two products of the clean-pair split."""
            l = P.l
            for gi, ((st_, gid), (x_src, rows, convs)) in enumerate(P.groups.items()):
                if st_ not in which:
                    continue
                if forked is not None and gi > 0:
                    # (slots 0 - 2; slot 3 is the direct conv's alone: it is joined later than these)
                    P.gmap.update(forked.run((gi - 1) % 3, lambda: self._stage_a(l, convs, x_src, rows[3], rows=rows[1], rows_cnt=rows[2])))
                else:
                    P.gmap.update(self._stage_a(l, convs, x_src, rows[3], rows=rows[1], rows_cnt=rows[2]))
            if forked is not None:
                for i_ in range(3):
                    forked.join(only=i_)
            c1 = P.c1
            if c1 is not None and "a" in which:      # atom<-atom at layer 1: touched edges per sample + the clean pairs once
                conv3 = m.conv_layers[9 * l + 3]
                rows_d, rows_dc = c1.rows_d, c1.rows_d_cnt
                if exact is not None:
                    rows_d, rows_dc = rows_d[:exact[rows_dc.data_ptr()]], None
                P.g_d = self._stage_a(l, [(3, conv3)], xa, Na, rows=rows_d, rows_cnt=rows_dc)
                P.x_clean = torch.empty((c1.n0, ldx), device=dev)
                K.gather_rows(xa, c1.rows_v, c1.n0, P.x_clean, ldx)
                P.g_v = self._stage_a(l, [(3, conv3)], P.x_clean, c1.n0)
                P.keep += [P.g_d, P.g_v, P.x_clean]

        def direct_tasks(P):
            """The direct convs (receptor<-atom: one edge per atom, nothing to factorise): no stage A."""
            l, spec = P.l, P.spec
            tasks, nb_d = [], 0.0
            # (the launch holds at most DDP_MAX_TASKS tasks: several direct convs in it - factorize_min_degree = 0 - share them)
            n_direct = sum(1 for k, (csr, so_k, x_src) in P.per.items() if not (so_k is not None or (k == 3 and P.c1 is not None)) and csr.n_edges > 0)
            split_cap = max(1, L.DDP_MAX_TASKS // max(1, n_direct))
            for k, (csr, so_k, x_src) in P.per.items():
                if so_k is not None or (k == 3 and P.c1 is not None):
                    continue
                x_recv, _, ek = arr[k]
                pkc = m.conv_layers[9 * l + k].packed(dev)
                msg = torch.empty((csr.n_edges, spec.d_out), device=dev)
                P.msgs[k] = (msg, csr, pkc)
                if csr.n_edges == 0:
                    continue
                if prof_on:
                    nb_d += node_bytes(l, k)
                segs = [(e[ek], csr.eid, ns, ns), (x_recv, csr.recv, ldx, ns), (x_src, csr.src, ldx, ns)]
                # (model.direct_rows: through the row-stationary kernel - 128-edge workgroups stream the fc.3 tiles once per 128 edges)
                # one conv as several tasks of segment ranges where its 128-edge workgroups would leave most of the chip empty (measured: 44
                # workgroups of 340 tiles - the 5-sample shard - 0.36 -> 0.155 ms per launch as six ranges; 347 workgroups - 40 samples - gain
                # nothing from two ranges, profiles/r06_direct_rows_ab.txt): ceil(256 / workgroups) ranges, at most direct_rows_max_split
                nsplit = max(1, min(m.direct_rows_max_split, split_cap, -(-256 // max(1, -(-csr.n_edges // 128)))))
                pkr = m.conv_layers[9 * l + k].packed_rows_direct(dev, nsplit) if nsplit > 1 else m.conv_layers[9 * l + k].packed_rows_direct(dev)
                if pkr is not None and all(K.rows_mode(p_) for p_ in (pkr if nsplit > 1 else [pkr])):
                    for i_, p_ in enumerate(pkr if nsplit > 1 else [pkr]):
                        t_ = K.make_task(p_, x_src, ldx, csr, sh[ek], segs, msg, rows=True)
                        if i_:
                            t_._count = (0, None)      # (the profiler counts a conv's edges once)
                        tasks.append(t_)
                else:
                    tasks.append(K.make_task(pkc, x_src, ldx, csr, sh[ek], segs, msg))
            P.tasks, P.nb_d = tasks, nb_d
            return tasks

        def launch_direct(P):
            K.launch_convs(P.spec, P.tasks, node_bytes=P.nb_d, tag=f"layer{P.l}")

        def launch_factorised(P, which="lar"):
            """The conv launch of the layer's factorised convs whose SOURCE-node type is in `which` (and that have not been launched yet)."""
            l, spec, c1 = P.l, P.spec, P.c1
            tasks_g, nb_g = [], 0.0
            for k, (csr, so_k, x_src) in P.per.items():
                if k in P.msgs or SRC_TYPE[k] not in which:
                    continue
                x_recv, _, ek = arr[k]
                conv = m.conv_layers[9 * l + k]
                pkc = conv.packed(dev)
                e_base, sh_k = e[ek], sh[ek]
                if k == 3 and c1 is not None:
                    # rows [0, E): per-sample messages at their CSR positions (only the touched edges are written and read),
                    # rows [E, E + e0): the messages of the complex's own edge list between clean atoms
                    msg = torch.empty((c1.E + c1.e0, spec.d_out), device=dev)
                    P.msgs[k] = (msg, csr, pkc, c1.rowmap)
                    pkg = conv.packed_g(dev)
                    rk = K.rows_mode(pkg)
                    sd_ = c1.so_d
                    if sd_.n_edges > 0:
                        segs = [(e_base, sd_.eid, ns, ns), (x_recv, sd_.recv, ldx, ns), (xa, sd_.src, ldx, ns)]
                        tasks_g.append(K.make_task(pkg, xa, ldx, sd_, sh_k, segs, msg, g=[P.g_d.get((3, s_)) for s_ in (0, 1)], rows=rk))
                    sv = c1.so_v
                    segs = [(e_base, sv.eid, ns, ns), (P.x_clean, sv.recv, ldx, ns), (P.x_clean, sv.src, ldx, ns)]
                    tasks_g.append(K.make_task(pkg, P.x_clean, ldx, sv, sh_k, segs, msg, g=[P.g_v.get((3, s_)) for s_ in (0, 1)], rows=rk))
                    continue
                msg = torch.empty((csr.n_edges, spec.d_out), device=dev)
                P.msgs[k] = (msg, csr, pkc)
                if csr.n_edges == 0:
                    continue
                if prof_on:
                    nb_g += node_bytes(l, k)
                segs = [(e_base, so_k.eid, ns, ns), (x_recv, so_k.recv, ldx, ns), (x_src, so_k.src, ldx, ns)]
                pkg = conv.packed_g(dev)
                tasks_g.append(K.make_task(pkg, x_src, ldx, so_k, sh_k, segs, msg, g=[P.gmap.get((k, s_)) for s_ in (0, 1)], rows=K.rows_mode(pkg)))
            P.tasks_g = getattr(P, "tasks_g", []) + tasks_g
            if tasks_g:
                K.launch_convs(P.spec_g, tasks_g, flops_spec=spec, node_bytes=nb_g, tag=f"layer{l}")

        def fix_rowmaps(P):
            if P.l == 0 and F.flex0 is not None:
                # flexible side chains: the kept receivers' messages were computed on the pruned lists; the segmented mean walks
                # the FULL lists and reads an unmarked receiver's messages from its copy in sample 0 (row map)
                for k, rm in F.flex0.rowmaps.items():
                    if k in P.msgs:
                        P.msgs[k] = (P.msgs[k][0], c[k], P.msgs[k][2], rm)

        def means(P, types):
            """Segmented mean + BatchNorm + residual (:315-324) of the node types in `types`, the reference's summation order."""
            spec, shared = P.spec, P.shared
            for rt in types:
                if not P.active[rt]:
                    continue
                x, n = nodes[rt]
                own = [P.msgs[k] for k in ORDER[rt] if k not in shared]
                if own:
                    K.launch_reduce(x, ldx, n, spec.d_out, own, accumulate=True)
                com = [P.msgs[k] for k in ORDER[rt] if k in shared]
                if com:   # graph 0's update of the shared convs, added to every graph's copy of the node
                    n0 = shared[[k for k in ORDER[rt] if k in shared][0]][0]
                    K.launch_reduce(x, ldx, n0, spec.d_out, com, accumulate=True, n_rep=B, rep_stride=n0)

        if pipelined:
            # The direct conv of layer l needs x(l) only - not stage A, not the 32-edge launch: it is started as soon as the means of
            # layer l - 1 are queued, on a stream of its own, and has stage A(l) AND the 32-edge launch of layer l to finish beside
            # (joined before the means of layer l touch x: no snapshot of x_atom).  The window between two 32-edge launches is then
            # [mean{rec} -> stage A{rec rows}] | [mean{lig} -> stage A{lig rows}] | [mean{atom} -> stage A{atom rows}].
            P = plan(0)
            direct_tasks(P)
            if P.tasks:
                side.run(1, lambda P=P: launch_direct(P))
            stage_a(P, "lar")
            def atom_g_bytes(Pn):
                """Bytes of G that stage A writes for the atom-source rows of a layer (host-side capacities)."""
                tot = 0
                for (st_, _gid), (_x, rows_, convs_) in Pn.groups.items():
                    if st_ != "a":
                        continue
                    for _k, conv_ in convs_:
                        pk_ = conv_.packed_g(dev)
                        tot += rows_[3] * 4 * sum(ld_ for ld_ in (getattr(pk_, "gh_ld", None) or []) if ld_ is not None)
                return tot
            for l in range(L_):
                nxt = plan(l + 1) if l + 1 < L_ else None
                launch_factorised(P)        # (split: the atom-sourced convs - the others were launched beside stage A{atom rows})
                join_lists()
                fix_rowmaps(P)
                side.join(only=1)           # direct conv(l) is done: the means below update x in place
                side.join(only=0)           # (split: the early conv launch of this layer)
                ev_r, ev_l, ev_sr, ev_sl = (torch.cuda.Event() for _ in range(4))

                def rec_chain(P=P, nxt=nxt, ev_r=ev_r, ev_sr=ev_sr):
                    means(P, "r")
                    ev_r.record()
                    if nxt is not None:
                        stage_a(nxt, "r")
                        ev_sr.record()

                def lig_chain(P=P, nxt=nxt, ev_l=ev_l, ev_sl=ev_sl):
                    means(P, "l")
                    ev_l.record()
                    if nxt is not None:
                        stage_a(nxt, "l")
                        ev_sl.record()
                side.run(3, rec_chain)
                side.run(2, lig_chain)
                means(P, "a")
                if nxt is not None:
                    direct_tasks(nxt)
                    if nxt.tasks:     # after mean{atom} (forked from main here) and mean{rec} (the event)
                        def direct_next(nxt=nxt, ev_r=ev_r):
                            torch.cuda.current_stream(dev).wait_event(ev_r)
                            launch_direct(nxt)
                        side.run(1, direct_next)
                    # (worth a second launch only where the atom rows' product is long: 4.7 GB at 40 samples of cfg2; 20 samples, 2.4 GB, and the
                    # README's small model, 1.4 GB, lost time with it: model.split_rows_min_g_bytes)
                    if bool(getattr(m, "split_rows_launch", False)) and atom_g_bytes(nxt) >= m.split_rows_min_g_bytes:
                        # the convs with receptor / ligand sources need x(l + 1) of every node type (the three means) and their own G: they
                        # start beside stage A of the atom rows (the largest product of the layer) instead of behind it
                        def early(nxt=nxt, evs=(ev_r, ev_l, ev_sr, ev_sl)):
                            st_ = torch.cuda.current_stream(dev)
                            for ev in evs:
                                st_.wait_event(ev)
                            if getattr(m, "shape_early_rows", False):
                                # one workgroup per CU (one 256-register wave per SIMD): stage A of the atom rows, queued beside this launch,
                                # finds a wave slot and 78 KiB of LDS on every CU (measured: tools/overlap_ab.py, profiles/r06_overlap_ab.txt)
                                K.occupancy_shaping(82 * 1024, 0)
                                try:
                                    launch_factorised(nxt, "rl")
                                finally:
                                    K.occupancy_shaping(0, 0)
                                return
                            launch_factorised(nxt, "rl")
                        side.run(0, early)
                    stage_a(nxt, "a")
                side.join(only=2)
                side.join(only=3)
                F.keep.append((P.keep, P.per, P.msgs, P.tasks, P.tasks_g, P.gmap, ev_r))
                P = nxt
            side.join()
            mark("reduce")
            return

        for l in range(L_):
            P = plan(l)
            # Small batches: the direct convs start FIRST, on a stream of their own, beside stage A - their few long workgroups (one
            # per 64 atoms, 130 KB of LDS: they cannot share a CU with the 32-edge kernel's) then hold their CUs before that kernel's
            # thousands arrive, instead of waiting for CUs it has drained (measured at 5 samples: 0.3 ms alone, 0.7 - 1.0 ms launched
            # beside it)
            direct_tasks(P)
            direct_first = fork is not None and bool(P.tasks)
            if direct_first:
                fork.run(3, lambda: launch_direct(P))
            stage_a(P, "lar", forked=fork)
            mark("conv_prep")
            launch_factorised(P)
            join_lists()
            if direct_first:
                fork.join()
            else:
                launch_direct(P)
            mark("conv_launch")
            fix_rowmaps(P)
            if dbg is not None:   # every conv's own output = segmented mean + BatchNorm of its messages alone
                for k, ent in P.msgs.items():
                    n_k = nodes[RECV_TYPE[k]][1]
                    o_k = torch.zeros((n_k, P.spec.d_out), device=dev)
                    K.launch_reduce(o_k, P.spec.d_out, n_k, P.spec.d_out, [ent], accumulate=False)
                    dbg[f"conv_layers.{9 * l + k}"] = o_k
            if fork is not None and m.fork_small_means:
                # small batches: the three node types' segmented means are ~20-us launches on disjoint arrays: side by side
                fork.run(0, lambda: means(P, "l"))
                fork.run(1, lambda: means(P, "r"))
                means(P, "a")
                fork.join()
            else:
                means(P, "lar")
            mark("reduce")
            F.keep.append((P.keep, P.per, P.msgs, P.tasks, P.tasks_g, P.gmap))

    # ================================================================================================ heads
    def _heads(self, data, S, F, lig, rec, atom, dev, mark):
        m = self.m
        ns, L_, B, ldx = m.ns, m.num_conv_layers, S.B, m._ldx
        Nl, Na = S.Nl, S.Na
        xl, xa, lpos = F.xl, F.xa, F.lpos
        lay_l, lay_a, lbatch, abatch = S.lay_l, S.lay_a, S.lbatch, S.abatch
        tr_sigma, rot_sigma, tor_sigma, sc_sigma = F.sig
        if m.confidence_mode:   # (:329-353) mean of the scalar channels per graph -> MLP
            def scalars(x):
                return torch.cat([x[:, :ns], x[:, m._d_final - ns:m._d_final]], dim=1) if L_ >= 3 else x[:, :ns]

            def graph_mean(v, b):   # (dense per-graph sums in a fixed order - index_add_'s float atomics are not reproducible)
                lay = G.DenseLayout.build(b, B)
                return lay.dense(v, 0.0).sum(1) / lay.counts.clamp(min=1).unsqueeze(1)

            conf_in = lay_l.dense(scalars(xl), 0.0).sum(1) / lay_l.counts.clamp(min=1).unsqueeze(1)   # (deterministic order)
            if m.flexible_sidechains:
                if S.num_flex > 0:
                    fr = data["flexResidues"]
                    bonds = lay_a.starts[fr.batch.long()] + fr.edge_idx.t().long()
                    flex_atoms = torch.unique(bonds)
                    conf_in = torch.cat([conf_in, graph_mean(scalars(xa)[flex_atoms], abatch[flex_atoms])], dim=1)
                else:
                    conf_in = torch.cat([conf_in, torch.zeros_like(conf_in)], dim=1)
            return m.confidence_predictor(conf_in).squeeze(dim=-1)

        dd = m.distance_embed_dim
        lib = L.load()
        # ---- torsion heads (:386-434).  The three read-outs are independent chains of five small launches each (embedding ->
        # harmonics -> conv -> mean -> MLP): the torsion heads run on forked streams beside the tr / rot head (parallel branches
        # of the captured step)
        fork = self._fork(dev) if (m.concurrent_heads and m.before_layers is None and m.section_timer is None) else None
        heads = []
        if F.tor is not None:
            heads.append(lambda: self._torsion_head(F.tor, "final_edge_embedding", m.tor_bond_conv, "tor_final_layer", xl, lpos, dev,
                                                    "tor_bond_conv", tor_sigma))
        if F.sc is not None:
            heads.append(lambda: self._torsion_head(F.sc, "sidechain_final_edge_embedding", m.sc_tor_bond_conv, "sc_tor_final_layer", xa,
                                                    F.apos, dev, "sc_tor_bond_conv", sc_sigma))
        preds = [fork.run(i, fn) for i, fn in enumerate(heads)] if fork is not None else None
        # ---- translation / rotation head (:357-384): ligand atoms -> their graph's centre (ddp_step_prologue: summed in index
        # order - index_add_'s float atomics land in an order that depends on what else the device is doing, and one ulp in
        # the centre is one ulp in tr / rot)
        ar_l = G.iota32(Nl, dev)
        b32_l = G._batch32(lay_l, Nl)
        center = F.center
        pk = m._edge_pack("center_edge_embedding", slice(0, dd), dev)
        e_c, sh_c = K.edge_featurize(pk, m.center_distance_expansion, center, b32_l, lpos, ar_l, F.pre["center"], ar_l)
        c_c = m._cached("c_c", (lig.batch,), lambda: G.build_csr(b32_l, ar_l, B, presorted=True))
        fspec = m.final_conv.spec
        pkc = m.final_conv.packed(dev)
        msg = torch.empty((Nl, fspec.d_out), device=dev)
        seg_idx = c_c.src if m.fixed_center_conv else c_c.recv
        K.launch_convs(fspec, [K.make_task(pkc, xl, ldx, c_c, sh_c, [(e_c, c_c.eid, ns, ns), (xl, seg_idx, ldx, ns)], msg)], tag="head")
        gp = torch.empty((B, fspec.d_out), device=dev)      # (every row is written: accumulate = False)
        K.launch_reduce(gp, fspec.d_out, B, fspec.d_out, [(msg, c_c, pkc)], accumulate=False)
        if m.debug_conv_outputs is not None:
            m.debug_conv_outputs["final_conv"] = gp
        data.graph_sigma_emb = F.graph_emb
        # read-out MLPs on [|v|, sigma embedding], score norms (:362-384): one launch
        a = L.TrRotArgs()
        a.gp, a.ld_gp, a.n_graphs, a.ns, a.sd, a.graph_emb = K._p(gp), fspec.d_out, B, ns, F.graph_emb.shape[1], K._p(F.graph_emb)
        tr_pred, rot_pred = torch.empty((B, 3), device=dev), torch.empty((B, 3), device=dev)
        hw = m._head_weights(dev)
        for i, name in enumerate(("tr_final_layer", "rot_final_layer")):
            a.w1[i], a.b1[i], a.w2[i], a.b2[i] = (K._p(t) for t in hw[name])
        a.out[0], a.out[1] = K._p(tr_pred), K._p(rot_pred)
        if m.scale_by_sigma:
            a.sigma[0], a.sigma[1] = K._p(tr_sigma), K._p(rot_sigma)
            a.so3_table, a.so3_n = K._p(hw["so3"]), hw["so3"].shape[0]
            a.so3_lo, a.so3_span = hw["so3_lo"], hw["so3_span"]
        L.check(lib.ddp_trrot_head(C.byref(a), K.stream()), "ddp_trrot_head")
        mark("center_head")
        if fork is not None:
            fork.join()
        else:
            preds = [fn() for fn in heads]
        preds = list(preds)
        tor_pred = preds.pop(0) if F.tor is not None else torch.empty(0, device=dev)
        sc_pred = preds.pop(0) if F.sc is not None else torch.empty(0, device=dev)
        mark("tor_heads")
        F.keep.append((center, e_c, sh_c, msg, gp, a, hw))
        return tr_pred, rot_pred, tor_pred, sc_pred

    def _torsion_head(self, h, mlp_name, conv, final_name, x, pos, dev, name, sigma):
        """build_bond_conv_graph / build_sidechain_conv_graph (:586-636) on the searched bond-centre graph + FullTensorProduct
        + tor_bond_conv + final layer + torus score norm (:386-434).  An empty graph gives zero scores (the reference fails
        there).  Launches: edge embedding, edge harmonics + bond attributes, conv, segmented mean, read-out."""
        m = self.m
        lib = L.load()
        ns, ldx = m.ns, m._ldx
        st, csr = h.st, h.csr
        E, T = csr.n_edges, st.T
        pk = m._edge_pack(mlp_name, slice(0, m.distance_embed_dim), dev)
        pre = pk.b1.reshape(1, -1).contiguous()
        zero_idx = m._cached("zeros_" + name, (st.bonds,), lambda: torch.zeros(st.cap, device=dev, dtype=torch.int32))
        e_t, sh_e = K.edge_featurize(pk, m.lig_distance_expansion, h.bond_pos, csr.recv, pos, csr.src, pre, zero_idx[:E], n_edges=E, cnt=csr.cnt)
        tor_sh = torch.empty((E, 4), device=dev)
        bond_attr = torch.empty((T, ns), device=dev)      # (:399,423) x[b0, :ns] + x[b1, :ns]
        L.check(lib.ddp_torsion_sh(K._p(sh_e), K._p(h.bond_vec), K._p(csr.recv), E, K._p(csr.cnt), K._p(tor_sh), K._p(x), ldx, ns,
                                   K._p(st.flat32[:T]), K._p(st.flat32[T:]), T, K._p(bond_attr), K.stream()), "ddp_torsion_sh")
        if m.smooth_edges and E > 0:      # (:614,634 -> :401,426) the bond-centre graphs' edges are weighted too
            tor_sh.mul_(self._smooth_weight(h.bond_pos, csr.recv, pos, csr.src, m.lig_max_radius).unsqueeze(1))
        spec, pkc = conv.spec, conv.packed(dev)
        msg = torch.empty((E, spec.d_out), device=dev)
        segs = [(e_t, csr.eid, ns, ns), (x, csr.src, ldx, ns), (bond_attr, csr.recv, ns, ns)]
        if E > 0:
            K.launch_convs(spec, [K.make_task(pkc, x, ldx, csr, tor_sh, segs, msg)], tag="head")
        hsum = torch.empty((T, spec.d_out), device=dev)      # (every row is written: accumulate = False)
        K.launch_reduce(hsum, spec.d_out, T, spec.d_out, [(msg, csr, pkc)] if E > 0 else [], accumulate=False)
        if m.debug_conv_outputs is not None:
            m.debug_conv_outputs[name] = hsum
        hw = m._head_weights(dev)
        out = torch.empty(T, device=dev)
        a = L.TorArgs()
        a.h, a.ld_h, a.n_bonds, a.ns = K._p(hsum), spec.d_out, T, ns
        a.w1, a.w2 = (K._p(t) for t in hw[final_name])
        a.out = K._p(out)
        if m.scale_by_sigma:
            a.sigma, a.graph_of_bond = K._p(sigma), K._p(st.batch32)
            a.torus_table, a.torus_n, a.torus_lo, a.torus_span = K._p(hw["torus"]), hw["torus"].shape[0] - 1, hw["torus_lo"], hw["torus_span"]
        L.check(lib.ddp_tor_head(C.byref(a), K.stream()), "ddp_tor_head")
        h.keep = (e_t, sh_e, tor_sh, bond_attr, msg, hsum, a, hw)
        return out
