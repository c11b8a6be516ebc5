"""Micro-benchmark of ddp_stage_a against torch.mm / torch.bmm on the stage-A shapes (diagnostic)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffdock_pocket_amd import _lib as L  # noqa: E402
from diffdock_pocket_amd.packing import gh3_ld, gh_dest_table, gh_ld, split_bf16x3, split_h2  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    for name, N, nb, ncols in (("atom", 44440, 2, 12672), ("rec", 5560, 6, 12672), ("lig", 1480, 6, 12672), ("lig5", 185, 6, 12672)):
        k, ldx = 60, 180
        x = torch.randn(N, ldx, device=dev)
        w = torch.randn(nb, k, ncols, device=dev)
        ldo = (ncols + 31) // 32 * 32
        out = torch.empty(nb, N, ldo, device=dev)
        offs = (C.c_int32 * nb)(*[120 * (i % 2) for i in range(nb)])
        t_mine = timeit(lambda: L.check(lib.ddp_stage_a(x.data_ptr(), ldx, N, None, None, N, offs, nb, w.data_ptr(), None, k, ncols, out.data_ptr(), ldo, st), "a"))
        w3 = split_bf16x3(w)
        t_x3 = timeit(lambda: L.check(lib.ddp_stage_a(x.data_ptr(), ldx, N, None, None, N, offs, nb, w.data_ptr(), w3.data_ptr(), k, ncols, out.data_ptr(), ldo, st), "a"))
        wh = split_h2(w)
        t_h2 = timeit(lambda: L.check(lib.ddp_stage_a_h2(x.data_ptr(), ldx, N, None, None, N, offs, nb, w.data_ptr(), wh.data_ptr(), k, ncols, out.data_ptr(), ldo, None, st), "a"))
        # the plane form ddp_conv_rows reads (ddp_stage_a_gh): a slot of 72 padded G columns (parts 32, 28, 12) at hid = 180
        ncg = gh_ld(180, 72)
        wg = torch.randn(nb, k, ncg, device=dev)
        wgh = split_h2(wg, unified_scale=0.5)
        outg = torch.empty(nb, N, ncg, device=dev)
        dest = torch.stack([gh_dest_table([32, 28, 12], 23, ncg)] * nb).contiguous().to(dev)
        t_gh = timeit(lambda: L.check(lib.ddp_stage_a_gh(x.data_ptr(), ldx, N, None, None, N, offs, nb, wg.data_ptr(), wgh.data_ptr(), k, ncg, outg.data_ptr(), ncg,
                                                         None, dest.data_ptr(), st), "a"))
        gbg = nb * N * ncg * 4 / 1e9
        print(f"{name}: plane form (ddp_stage_a_gh) {gbg:.2f} GB out: {t_gh:.3f} ms ({gbg / t_gh:.2f} TB/s)")
        # plane form 1 (fp16 hi + continuation bytes, ddp_stage_a_gh3): 24 bytes per 8 values
        ld3 = gh3_ld(180, 72)
        nc3 = ld3 // 6 * 8
        wg3 = torch.randn(nb, k, nc3, device=dev)
        wgh3 = split_h2(wg3, unified_scale=0.5)
        outg3 = torch.empty(nb, N, ld3, device=dev)
        dest3 = torch.stack([gh_dest_table([32, 28, 12], 23, nc3, fmt=1)] * nb).contiguous().to(dev)
        t_g3 = timeit(lambda: L.check(lib.ddp_stage_a_gh3(x.data_ptr(), ldx, N, None, None, N, offs, nb, wg3.data_ptr(), wgh3.data_ptr(), k, nc3, outg3.data_ptr(), ld3,
                                                          None, dest3.data_ptr(), st), "a"))
        gb3 = nb * N * ld3 * 4 / 1e9
        print(f"{name}: plane form 1 (ddp_stage_a_gh3) {gb3:.2f} GB out: {t_g3:.3f} ms ({gb3 / t_g3:.2f} TB/s) = x{t_g3 / t_gh:.2f} of the 4-byte form's time")
        A = torch.stack([x[:, 120 * (i % 2):120 * (i % 2) + k] for i in range(nb)])
        t_bmm = timeit(lambda: torch.bmm(A, w))
        t_mm = timeit(lambda: [torch.mm(A[i], w[i]) for i in range(nb)])
        gb = nb * N * ncols * 4 / 1e9
        print(f"{name}: N={N} nb={nb}  {gb:.2f} GB out | ddp_stage_a fp32 {t_mine:.3f} ms ({gb / t_mine:.2f} TB/s)  bf16x3 {t_x3:.3f} ms ({gb / t_x3:.2f})  fp16 hi/lo {t_h2:.3f} ms ({gb / t_h2:.2f})  bmm {t_bmm:.3f} ms ({gb / t_bmm:.2f})  "
              f"mm x{nb} {t_mm:.3f} ms ({gb / t_mm:.2f})")


if __name__ == "__main__":
    main()
