#!/usr/bin/env python3
"""OT side of the path on the device-resident API, 4 Mi OTs: IKNP expansion (receiver / sender), bit-COT, the COT pad
loops over MITCCRH (sender / receiver) and the KOS consistency check, as OT/s and algorithmic GB/s.  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mpc_amd import engine
from mpc_amd.circuit import LABEL, WIRE


def timed(ctx, fn, reps):
    # steady state: warm up for >= 2 calls and 40 ms (clocks ramp after host-side set-up), then time >= 10 calls
    t_warm, it = time.perf_counter(), 0
    while it < 2 or time.perf_counter() - t_warm < 0.04:
        fn()
        ctx.sync()
        it += 1
    reps = max(reps, 10)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


def run(n=1 << 22, reps=10, ctx=None):
    from scripts.bench_iknp import run as iknp_run
    own = ctx is None
    if own:
        ctx = engine.Context(0)
    res = {"n_ots": n, "iknp": iknp_run(n, reps, ctx=ctx)}
    rng = np.random.default_rng(11)
    lab = lambda: (int(rng.integers(0, 1 << 63)), int(rng.integers(0, 1 << 63)))
    seed, delta, seed2 = lab(), lab(), lab()
    d_data = ctx.random_u8((n, 16), 256, seed=1)
    d_wires = ctx.random_u8((n, 32), 256, seed=2)
    d_out = ctx.zeros((2 * n, 16))
    d_flags = ctx.random_u8((n,), 2, seed=3)
    d_res = ctx.random_u8((n, 16), 256, seed=4)
    ts = timed(ctx, lambda: engine.cot_send_pads_dev(ctx, seed, delta, d_data, d_wires, n, d_out), reps)
    tr = timed(ctx, lambda: engine.cot_receive_unpad_dev(ctx, seed, d_flags, d_out, d_res, n), reps)
    # per OT: sender reads 16 + 32 B, writes 32 B, 2 AES-128 blocks + 1 key schedule; receiver reads 16 + 16(+16) + 1 B, writes 16 B
    res["cot"] = {"send_ms": ts * 1e3, "recv_ms": tr * 1e3, "send_ot_per_s": n / ts, "recv_ot_per_s": n / tr,
                  "send_alg_GBs": 80 * n / ts / 1e9, "recv_alg_GBs": 49 * n / tr / 1e9,
                  "kernels": "classic" if os.environ.get("GC_COT_CLASSIC") else "dual-table persistent"}
    cv = np.zeros(256, LABEL)
    cv["d0"] = rng.integers(0, 1 << 63, 256, dtype=np.uint64)
    bcv = rng.integers(0, 2, 256).astype(np.uint8)
    tk = timed(ctx, lambda: engine.kos_receiver_tags_dev(ctx, seed2, d_res, d_flags, n, cv, bcv), max(2, reps // 3))
    res["kos"] = {"receiver_tags_ms": tk * 1e3, "ot_per_s": n / tk}
    # bit-COT
    base = np.zeros(128, WIRE)
    for f in ("l0", "l1"):
        base[f]["d0"] = rng.integers(0, 1 << 63, 128, dtype=np.uint64)
        base[f]["d1"] = rng.integers(0, 1 << 63, 128, dtype=np.uint64)
    rx = engine.IKNPReceiver(ctx, base)
    d_c = ctx.random_u8((((n + 63) // 64) * 8,), 256, seed=5)
    d_u = ctx.zeros(((n + 511) // 512) * 8192)
    d_r = ctx.zeros(((n + 63) // 64) * 8)
    tb = timed(ctx, lambda: rx.receive_bits_dev(d_c, n, d_u, d_r), reps)
    res["bitcot"] = {"receive_bits_ms": tb * 1e3, "ot_per_s": n / tb}
    rx.close()
    if own:
        ctx.close()
    return res


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22)))
