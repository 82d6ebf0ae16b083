#!/usr/bin/env python3
"""Cycle breakdown of the fused garble / eval kernels (s_memtime instrumentation, developer aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_amd import engine, parse_file

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
keylen = int(sys.argv[2]) if len(sys.argv) > 2 else 32
name = sys.argv[3] if len(sys.argv) > 3 else "aes_128.gcf"
c = parse_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name))
ctx = engine.Context(0)
dc = engine.DeviceCircuit(ctx, c)
gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
key = bytes(range(keylen))
d_rnd = ctx.random_u8((batch, c.num_inputs + 1, 16), 256, seed=1)
d_bits = ctx.random_u8((batch, c.num_inputs), 2, seed=2)
for _ in range(3):
    gb.garble(key, d_rnd); ev.select_inputs(gb, d_bits); ev.eval(key, gb)
ctx.sync()
print("plain: garble %.3f ms eval %.3f ms" % (gb.last_ms, ev.last_ms))
gb.debug_profile(True); ev.debug_profile(True)
gb.garble(key, d_rnd); ev.select_inputs(gb, d_bits); ev.eval(key, gb)
ctx.sync()
print("instrumented: garble %.3f ms eval %.3f ms" % (gb.last_ms, ev.last_ms))
names = ["header", "hash-post", "barA", "commit", "xor", "barB", "hash-pre", "aes"]
for nm, b in (("garble", gb), ("eval", ev)):
    p = b.debug_profile(True, read=True)
    tot0, tot1 = sum(p[:8]), sum(p[8:])
    print(nm, "wave0:", {n: int(v) for n, v in zip(names, p[:8])}, "total", int(tot0))
    print(nm, "wave15:", {n: int(v) for n, v in zip(names, p[8:16])}, "total", int(tot1))
