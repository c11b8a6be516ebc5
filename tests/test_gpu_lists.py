"""The batched index-list primitives with device-side counts (csrc/ddp_lists.hip, ddp_graph.hip, ddp_views.hip) against their
PyTorch definitions: ragged inputs, counts below the capacity, empty lists, several jobs per launch."""
import pytest
import torch

from diffdock_pocket_amd import graph as G
from diffdock_pocket_amd import launch as K

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _i32(t, dev):
    return t.to(torch.int32).contiguous().to(dev)


def test_scan_jobs_flags_counts_and_row_lengths():
    dev = _dev()
    g = torch.Generator().manual_seed(0)
    jobs, checks = [], []
    for n, cap, base in ((0, 5, 3), (1, 1, 0), (63, 64, 0), (64, 64, 7), (1000, 1500, 0), (40000, 44440, 11)):
        flag = (torch.rand(cap, generator=g) < 0.3).int()
        val = torch.randint(0, 5, (cap,), generator=g).int()
        rowptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(torch.randint(0, 9, (cap,), generator=g), 0)]).int()
        n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
        for mode in ("flag_list", "val", "flag_rowptr"):
            fd, vd, rd = _i32(flag, dev), _i32(val, dev), _i32(rowptr, dev)
            excl = torch.full((cap + 1,), -7, dtype=torch.int32, device=dev)
            excl2 = torch.full((cap + 1,), -7, dtype=torch.int32, device=dev)
            lst = torch.full((cap,), -7, dtype=torch.int32, device=dev)
            total = torch.full((1,), -7, dtype=torch.int32, device=dev)
            if mode == "flag_list":
                jobs.append(K.scan_job(cap, flag=fd, base=base, excl=excl, lst=lst, total=total, n_dev=n_dev))
                v = flag[:n].long()
            elif mode == "val":
                jobs.append(K.scan_job(cap, val=vd, base=base, excl=excl, excl2=excl2, total=total, n_dev=n_dev))
                v = val[:n].long()
            else:
                jobs.append(K.scan_job(cap, flag=fd, rowptr=rd, base=base, excl=excl, total=total, n_dev=n_dev))
                v = (flag[:n].long() * (rowptr[1:n + 1] - rowptr[:n]).long())
            want = base + torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(v, 0)])
            checks.append((mode, n, excl, excl2, lst, total, want, flag[:n], (fd, vd, rd, n_dev)))
    K.scan_jobs(jobs)      # more than DDP_MAX_LIST_JOBS: split over launches
    torch.cuda.synchronize()
    for mode, n, excl, excl2, lst, total, want, flag, _ in checks:
        assert torch.equal(excl[:n + 1].cpu().long(), want), (mode, n)
        assert int(total.item()) == int(want[-1])
        assert bool((excl[n + 1:] == -7).all())
        if mode == "val":
            assert torch.equal(excl2[:n].cpu().long(), want[:n])
        if mode == "flag_list":
            idx = flag.nonzero(as_tuple=True)[0]
            k = idx.shape[0]
            # (list entries are written at excl[i], which includes the base)
            base = int(want[0])
            assert torch.equal(lst[base:base + k].cpu().long(), idx) if base + k <= lst.shape[0] else True


def test_mark_rowcopy_select_gather():
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    N, E = 500, 20000
    recv = torch.sort(torch.randint(0, N, (E,), generator=g)).values
    src = torch.randint(0, N, (E,), generator=g)
    eid = torch.randperm(E, generator=g)
    rowptr = torch.zeros(N + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(recv, minlength=N), 0)
    keep = (torch.rand(N, generator=g) < 0.4).int()
    n_act = 15000                                    # device-side count below the capacity
    n_dev = torch.tensor([n_act], dtype=torch.int32, device=dev)
    rd, sd, ed, rpd, kd = (_i32(t, dev) for t in (recv, src, eid, rowptr, keep))
    # mark
    mask = torch.zeros(N, dtype=torch.int32, device=dev)
    K.mark_jobs([K.mark_job(mask, sd, E, n_dev)])
    want_mask = torch.zeros(N, dtype=torch.int32)
    want_mask[src[:n_act]] = 1
    assert torch.equal(mask.cpu(), want_mask)
    # row compaction (full list)
    new_rp = torch.empty(N + 1, dtype=torch.int32, device=dev)
    tot = torch.empty(1, dtype=torch.int32, device=dev)
    K.scan_jobs([K.scan_job(N, flag=kd, rowptr=rpd, excl=new_rp, total=tot)])
    o = [torch.full((E,), -1, dtype=torch.int32, device=dev) for _ in range(3)]
    K.rowcopy_jobs([K.rowcopy_job(N, kd, rpd, new_rp, [rd, sd, ed], o)])
    sel = keep[recv].bool()
    n_keep = int(sel.sum())
    assert int(tot.item()) == n_keep
    for got, ref in zip(o, (recv, src, eid)):
        assert torch.equal(got[:n_keep].cpu().long(), ref[sel])
        assert bool((got[n_keep:] == -1).all())
    # general selection: items with a flagged end, payload gathers, device-side count
    touched = (torch.rand(N, generator=g) < 0.2).int()
    td = _i32(touched, dev)
    outs = [torch.full((E,), -1, dtype=torch.int32, device=dev) for _ in range(4)]
    oidx = torch.full((E,), -1, dtype=torch.int32, device=dev)
    tot2 = torch.empty(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(2 * ((E + 2047) // 2048) + 1, dtype=torch.int32, device=dev)
    empty_tot = torch.full((1,), 9, dtype=torch.int32, device=dev)
    K.select_jobs([K.select_job(E, td, rd, td, sd, [rd, sd, ed, ed], outs, tot2, scratch, out_idx=oidx, n_dev=n_dev, pay_add=[0, 0, 0, 5]),
                   K.select_job(0, td, rd, None, None, [], [], empty_tot, scratch)])
    f = (touched[recv].bool() | touched[src].bool())
    f[n_act:] = False
    k = int(f.sum())
    assert int(tot2.item()) == k and int(empty_tot.item()) == 0
    assert torch.equal(oidx[:k].cpu().long(), f.nonzero(as_tuple=True)[0])
    for got, ref, add in zip(outs, (recv, src, eid, eid), (0, 0, 0, 5)):
        assert torch.equal(got[:k].cpu().long(), ref[f] + add)
    # row gather
    x = torch.randn(N, 37, generator=g)
    idx = torch.randint(0, N, (300,), generator=g)
    out = torch.zeros(300, 40, device=dev)
    K.gather_rows(x.to(dev), _i32(idx, dev), 300, out, 37, n_dev=torch.tensor([250], dtype=torch.int32, device=dev))
    assert torch.equal(out[:250, :37].cpu(), x[idx[:250]]) and float(out[250:].abs().max()) == 0.0 and float(out[:, 37:].abs().max()) == 0.0


def test_group_by_key_jobs_with_device_counts():
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    jobs, checks = [], []
    for E, n_act, n_keys in ((5000, 3000, 70), (64, 64, 200), (10, 0, 5), (20000, 20000, 1500)):
        key = torch.randint(0, n_keys, (E,), generator=g)
        p0, p1 = torch.randint(0, 1000, (E,), generator=g), torch.randint(0, 1000, (E,), generator=g)
        kmap = torch.randperm(n_keys, generator=g)
        kd, p0d, p1d, kmd = (_i32(t, dev) for t in (key, p0, p1, kmap))
        n_dev = torch.tensor([n_act], dtype=torch.int32, device=dev)
        rp = torch.empty(n_keys + 1, dtype=torch.int32, device=dev)
        perm, ok, o0, o1 = (torch.full((E,), -1, dtype=torch.int32, device=dev) for _ in range(4))
        scratch = torch.empty(n_keys + E, dtype=torch.int32, device=dev)
        jobs.append(K.group_job(kd, E, n_keys, [p0d, p1d], rp, perm, ok, [o0, o1], scratch, n_dev=n_dev, key_map=kmd))
        order = torch.sort(key[:n_act], stable=True).indices
        want_rp = torch.zeros(n_keys + 1, dtype=torch.long)
        want_rp[1:] = torch.cumsum(torch.bincount(key[:n_act], minlength=n_keys), 0)
        checks.append((n_act, rp, perm, ok, o0, o1, want_rp, order, kmap[key[:n_act]][order], p0[:n_act][order], p1[:n_act][order],
                       (kd, p0d, p1d, kmd, n_dev, scratch)))
        # rowptr only (perm == NULL)
        rp2 = torch.empty(n_keys + 1, dtype=torch.int32, device=dev)
        sc2 = torch.empty(n_keys, dtype=torch.int32, device=dev)
        jobs.append(K.group_job(kd, E, n_keys, [], rp2, scratch=sc2, n_dev=n_dev))
        checks.append((None, rp2, want_rp, sc2))
    K.group_jobs(jobs)
    torch.cuda.synchronize()
    for ch in checks:
        if ch[0] is None:
            assert torch.equal(ch[1].cpu().long(), ch[2])
            continue
        n_act, rp, perm, ok, o0, o1, want_rp, order, wkey, w0, w1, _ = ch
        assert torch.equal(rp.cpu().long(), want_rp)
        assert torch.equal(perm[:n_act].cpu().long(), order)
        assert torch.equal(ok[:n_act].cpu().long(), wkey) and torch.equal(o0[:n_act].cpu().long(), w0) and torch.equal(o1[:n_act].cpu().long(), w1)
        assert bool((perm[n_act:] == -1).all())


@pytest.mark.parametrize("scaled", [False, True])
def test_radius_search_jobs_match_the_dense_definition(scaled):
    """count -> scan -> fill without the host, several searches per launch, a base offset, a count-only job, the per-graph
    scaled search of the dynamic cross cutoff - against graph.radius / radius_graph on the CPU (dense formulation)."""
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    sizes_x, sizes_y = [30, 0, 55, 17], [12, 9, 0, 21]
    x = torch.cat([torch.randn(n, 3, generator=g) * 3 + 2 * i for i, n in enumerate(sizes_x)])
    y = torch.cat([torch.randn(n, 3, generator=g) * 3 + 2 * i for i, n in enumerate(sizes_y)])
    bx = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(sizes_x)])
    by = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(sizes_y)])
    B = 4
    lx, ly = G.DenseLayout.build(bx, B), G.DenseLayout.build(by, B)
    div = torch.tensor([2.0, 3.0, 1.5, 2.5])
    r = 1.4 if scaled else 3.0
    xs, ys = (x / div[bx].unsqueeze(1), y / div[by].unsqueeze(1)) if scaled else (x, y)
    want = G.radius(xs, ys, r, lx, ly, max_num_neighbors=10000)              # [query; x]
    want_g = G.radius_graph(xs, r, lx, max_num_neighbors=6)                  # [neighbour; query]
    lxd, lyd = G.DenseLayout.build(bx.to(dev), B), G.DenseLayout.build(by.to(dev), B)
    G._ptr(lxd), G._ptr(lyd)
    xd, yd, dd = x.to(dev), y.to(dev), (div.to(dev) if scaled else None)
    i32e = lambda n: torch.full((n,), -1, dtype=torch.int32, device=dev)      # noqa: E731
    ny, nx = y.shape[0], x.shape[0]
    cap, base = nx * 12, 5
    oq, ox, tot = i32e(cap), i32e(cap), i32e(1)
    gq, gx, gtot = i32e(nx * 7 + base), i32e(nx * 7 + base), i32e(1)
    cnt_only = i32e(ny)
    jobs = [K.radius_job(xd, lxd._ptr32, yd, G._batch32(lyd, ny), r, 10000, 0, i32e(ny), i32e(ny + 1), total=tot, out_query=oq, out_x=ox,
                         capacity=cap, graph_div=dd),
            K.radius_job(xd, lxd._ptr32, xd, G._batch32(lxd, nx), r, 7, 1, i32e(nx), i32e(nx + 1), base=base, total=gtot, out_query=gq,
                         out_x=gx, capacity=nx * 7 + base, graph_div=dd),
            K.radius_job(xd, lxd._ptr32, yd, G._batch32(lyd, ny), r, 10000, 0, cnt_only, graph_div=dd)]
    K.radius_search_jobs(jobs)
    torch.cuda.synchronize()
    E = want.shape[1]
    assert int(tot.item()) == E
    assert torch.equal(oq[:E].cpu().long(), want[0]) and torch.equal(ox[:E].cpu().long(), want[1])
    Eg = want_g.shape[1]
    assert int(gtot.item()) == Eg + base
    assert torch.equal(gx[base:base + Eg].cpu().long(), want_g[0]) and torch.equal(gq[base:base + Eg].cpu().long(), want_g[1])
    assert bool((gq[:base] == -1).all())
    assert torch.equal(cnt_only.cpu().long(), torch.bincount(want[0], minlength=ny))


def test_clean_pair_maps():
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    B, n0, e0 = 5, 40, 130
    recv0 = torch.sort(torch.randint(0, n0, (e0,), generator=g)).values
    src0 = torch.randint(0, n0, (e0,), generator=g)
    recv = torch.cat([recv0 + b * n0 for b in range(B)])
    src = torch.cat([src0 + b * n0 for b in range(B)])
    touched = (torch.rand(B * n0, generator=g) < 0.3).int()
    touched[3::n0] = 1                                           # atom 3 is touched in every sample
    E = B * e0
    rowmap = torch.empty(E, dtype=torch.int32, device=dev)
    rows_v = torch.empty(n0, dtype=torch.int32, device=dev)
    K.clean_pair_maps(_i32(touched, dev), _i32(recv, dev), _i32(src, dev), E, e0, B, n0, rowmap, rows_v)
    dirty = touched[recv].bool() | touched[src].bool()
    p = torch.arange(E)
    assert torch.equal(rowmap.cpu().long(), torch.where(dirty, p, E + p % e0))
    clean = (touched.view(B, n0) == 0)
    first = clean.to(torch.uint8).argmax(0)
    assert torch.equal(rows_v.cpu().long(), first * n0 + torch.arange(n0))


@pytest.mark.parametrize("a_too,ref_list,by_flag", [(True, True, False), (False, False, False), (True, True, True), (False, True, True)])
def test_flex_mark_and_fallback_rowmap(a_too, ref_list, by_flag):
    """ddp_flex_mark / ddp_fallback_rowmap (partial sharing with moving atoms) against their definitions: B samples of n nodes,
    e0 edges each stored sample after sample; "off" = position differs from sample 0's, or a given flag."""
    dev = _dev()
    torch.manual_seed(3 + a_too + 2 * by_flag)
    B, n, deg = 5, 57, 4
    e0 = n * deg
    pos0 = torch.randn(n, 3)
    pos = pos0.repeat(B, 1, 1)
    moved = torch.rand(B, n) < 0.1
    moved[0] = False
    pos[moved] += 0.25
    flag = (torch.rand(B, n) < 0.15).int()
    off = (flag != 0) if by_flag else moved
    a_loc = torch.stack([torch.randint(0, n, (e0,)).sort().values for _ in range(B)])       # per-sample lists, node-major
    b_loc = torch.randint(0, n, (B, e0))
    offs = (torch.arange(B) * n).unsqueeze(1)
    a, b = (a_loc + offs).reshape(-1).int(), (b_loc + offs).reshape(-1).int()
    want = torch.zeros(B, n, dtype=torch.int32)
    want[0] = 1
    for s in range(1, B):
        hit = off[s][b_loc[s]] | (off[s][a_loc[s]] if a_too else False)
        want[s][a_loc[s][hit]] = 1
        if ref_list:
            want[s][a_loc[0][off[s][b_loc[0]]]] = 1
    mark = torch.zeros(B * n, dtype=torch.int32, device=dev)
    K.flex_mark(a.to(dev), b.to(dev), B * e0, e0, n, n, mark, pos=None if by_flag else pos.reshape(-1, 3).to(dev),
                flag=flag.reshape(-1).to(dev) if by_flag else None, a_too=a_too, ref_list=ref_list)
    assert torch.equal(mark.cpu().reshape(B, n), want)
    # fallback row map: kept rows = marked receivers; an unmarked receiver reads its copy's rows in sample 0 (equal row lengths there)
    a_same = (a_loc[0].unsqueeze(0) + offs).reshape(-1)                                        # every sample with sample 0's rows
    rowptr = torch.zeros(B * n + 1, dtype=torch.long)
    rowptr[1:] = torch.bincount(a_same, minlength=B * n).cumsum(0)
    keep = want.reshape(-1).bool()
    lens = (rowptr[1:] - rowptr[:-1]) * keep
    new_rowptr = torch.zeros(B * n + 1, dtype=torch.long)
    new_rowptr[1:] = lens.cumsum(0)
    p = torch.arange(B * e0)
    r = a_same
    t = torch.where(keep[r], r, r % n)
    want_map = new_rowptr[t] + (p - rowptr[r])
    rowmap = torch.empty(B * e0, dtype=torch.int32, device=dev)
    K.fallback_rowmap(want.reshape(-1).to(dev), a_same.int().to(dev), rowptr.int().to(dev), new_rowptr.int().to(dev), B * e0, n, rowmap)
    assert torch.equal(rowmap.cpu().long(), want_map)
