#!/usr/bin/env python3
"""Cycle breakdown (s_memtime instrumentation) of the kernels that run a synthetic wide circuit (developer aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_amd import engine
from mpc_amd.circuit import synthetic_levelised

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
f = float(sys.argv[2]) if len(sys.argv) > 2 else 0.17
batch = 1024
c = synthetic_levelised(max(2, 131072 // W), W, f, seed=105, ninputs=256)
ctx = engine.Context(0)
dc = engine.DeviceCircuit(ctx, c)
gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
key = bytes(range(32))
d_rnd = ctx.random_u8((batch, c.num_inputs + 1, 16), 256, seed=1)
d_bits = ctx.random_u8((batch, c.num_inputs), 2, seed=2)
for _ in range(3):
    gb.garble(key, d_rnd); ev.select_inputs(gb, d_bits); ev.eval(key, gb)
ctx.sync()
print("lds_wires", gb.lds_wires, "tile", gb.tile_instances, "plain: garble %.3f ms eval %.3f ms" % (gb.last_ms, ev.last_ms))
gb.debug_profile(True); ev.debug_profile(True)
gb.garble(key, d_rnd); ev.select_inputs(gb, d_bits); ev.eval(key, gb)
ctx.sync()
print("instrumented: garble %.3f ms eval %.3f ms" % (gb.last_ms, ev.last_ms))
for nm, b in (("garble", gb), ("eval", ev)):
    p = b.debug_profile(True, read=True)
    print(nm, "wave0 :", [int(v) for v in p[:8]], "total", int(sum(p[:8])))
    print(nm, "waveN :", [int(v) for v in p[8:16]], "total", int(sum(p[8:16])))
