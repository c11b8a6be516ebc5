import torch
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for gb in (0.45, 1.68, 4.48):
    n = int(gb * 1e9 / 4)
    out = torch.empty(n, device=dev)
    src = torch.randn(n, device=dev)
    t = timeit(lambda: out.fill_(1.0))
    t2 = timeit(lambda: out.copy_(src))
    t3 = timeit(lambda: torch.mul(src, 2.0, out=out))
    print(f"{gb} GB: fill {t:.3f} ms ({gb/t:.2f} TB/s written)  copy {t2:.3f} ms ({gb/t2:.2f} TB/s written, {2*gb/t2:.2f} total)  mul {t3:.3f} ms ({gb/t3:.2f})")
