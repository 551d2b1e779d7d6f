#!/usr/bin/env python3
"""Development aid: what this box sustains for a plain device copy / read (torch ops), as the practical HBM ceiling
to compare the bandwidth-bound kernels with."""
import torch, time
dev = torch.device("cuda", 0)
for mb in (512, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev).normal_(); y = torch.empty_like(x)
    for name, fn, bytes_ in (("copy (read+write)", lambda: y.copy_(x), 8 * n), ("sum (read)", lambda: x.sum(), 4 * n),
                             ("add (2 reads + write)", lambda: torch.add(x, y, out=y), 12 * n)):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        print(f"{mb:5d} MB {name:22s}: {bytes_ / dt / 1e12:.2f} TB/s")
