#!/usr/bin/env python3
"""What can this box's host link do?  Pinned-memory DMA (hipMemcpyAsync through torch), each direction alone and both at
once, and a device kernel writing / reading pinned host memory directly (zero-copy).  Reference for bench_host_api.py."""
import json
import time

import torch

n = 256 << 20
h_out = torch.empty(n, dtype=torch.uint8, pin_memory=True)
h_in = torch.empty(n, dtype=torch.uint8, pin_memory=True)
h_in.fill_(3)
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def d2h():
    with torch.cuda.stream(s1):
        h_out.copy_(d_a, non_blocking=True)


def h2d():
    with torch.cuda.stream(s2):
        d_b.copy_(h_in, non_blocking=True)


def both():
    d2h()
    h2d()


res = {"bytes": n}
res["d2h_GBs"] = n / timed(d2h) / 1e9
res["h2d_GBs"] = n / timed(h2d) / 1e9
t = timed(both)
res["duplex_GBs_each"] = n / t / 1e9
print(json.dumps(res))
