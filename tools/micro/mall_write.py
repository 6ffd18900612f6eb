#!/usr/bin/env python3
"""Does the 256 MB Infinity Cache absorb REPEATED writes of a buffer that fits it?  (question behind: could the all-angle forward
projector's partial sums live there if a pass wrote <= ~150 MB of them)  torch fill_ / copy_ as the probes."""
import time
import torch
dev = torch.device("cuda", 0)
for mb in (32, 64, 96, 128, 160, 192, 256, 384, 512, 1024):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    for name, fn, nbytes in (("fill", lambda: a.fill_(1.0), mb << 20), ("copy a->b", lambda: b.copy_(a), 2 * (mb << 20)),
                             ("fill a then read a (sum)", lambda: (a.fill_(2.0), a.sum()), 2 * (mb << 20))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 30
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{mb:5d} MB  {name:28s} {dt * 1e6:8.1f} us  {nbytes / dt / 1e12:6.2f} TB/s")
