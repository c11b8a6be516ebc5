"""Diagnostic (DDP_SA_TUNE builds of the library): ddp_stage_a_h2 on the products below 8192 rows - the 5-sample shard's and the
40-sample batch's receptor / ligand rows - against the rows per workgroup (DDP_SA_MROWS) and, for listed products, the grid cap."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffdock_pocket_amd import _lib as L  # noqa: E402
from diffdock_pocket_amd.packing import split_h2  # noqa: E402
from bench_stage_a import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    k, ldx, ncols = 60, 180, 12672
    ldo = (ncols + 31) // 32 * 32
    for name, N, nb, n_list in (("atom5", 5555, 2, 0), ("rec5", 695, 6, 0), ("lig5", 185, 6, 0), ("near5", 5555, 2, 363),
                                ("rec40", 5560, 6, 0), ("lig40", 1480, 6, 0), ("near40", 44440, 2, 2903), ("pruned40", 44440, 2, 12000)):
        x = torch.randn(N, ldx, device=dev)
        w = torch.randn(nb, k, ncols, device=dev)
        wh = split_h2(w)
        out = torch.empty(nb, N, ldo, device=dev)
        offs = (C.c_int32 * nb)(*[120 * (i % 2) for i in range(nb)])
        rows = cnt = None
        n = N
        if n_list:
            rows = torch.randperm(N, device=dev)[:N].sort().values.to(torch.int32).contiguous()
            cnt = torch.tensor([n_list], dtype=torch.int32, device=dev)
        line = []
        for mrows in (32, 64, 128, 256, 512):
            for cap in ((96, 256, 1024) if n_list else (96,)):
                os.environ["DDP_SA_MROWS"], os.environ["DDP_SA_GYCAP"] = str(mrows), str(cap)
                t = timeit(lambda: L.check(lib.ddp_stage_a_h2(x.data_ptr(), ldx, N, rows.data_ptr() if rows is not None else None,
                                                              cnt.data_ptr() if cnt is not None else None, N, offs, nb, w.data_ptr(),
                                                              wh.data_ptr(), k, ncols, out.data_ptr(), ldo, None, st), "a"), n=20)
                line.append(f"{mrows}{'/' + str(cap) if n_list else ''}: {t * 1e3:.1f}")
        gb = nb * (n_list or N) * ncols * 4 / 1e9
        print(f"{name}: rows {n_list or N} of {N}, nb {nb}, {gb:.3f} GB | us by rows per workgroup{' / grid cap' if n_list else ''}: " + "  ".join(line), flush=True)


if __name__ == "__main__":
    main()
