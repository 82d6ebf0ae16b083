#!/usr/bin/env python3
"""SURVEY.md §8d synthetic levelised circuits: generator(levels, width, AND fraction, seed), gate at level l
picks its inputs from levels < l; W in {64, 1024, 16384}, f in {0, 0.17, 0.5, 1}; plus buildANDChain(n)
(circuit/garble_bench_test.go:19) as the worst case (depth n, width 1).  One JSON object per line:
AND-gates/s and gates/s (garble+eval, device-resident), which kernels ran, and an output check against
plaintext evaluation."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mpc_amd import engine
from mpc_amd.circuit import and_chain, synthetic_levelised

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
gates_target = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
key = bytes(range(32))
ctx = engine.Context(0)
cases = [(w, f) for w in (64, 1024, 16384) for f in (0.0, 0.17, 0.5, 1.0)]
circs = [synthetic_levelised(max(2, gates_target // w), w, f, seed=100 + i, ninputs=256) for i, (w, f) in enumerate(cases)]
circs.append(and_chain(4096))
for c in circs:
    dc = engine.DeviceCircuit(ctx, c)
    info = dc.info
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    d_rnd = torch.randint(0, 256, (batch, c.num_inputs + 1, 16), dtype=torch.uint8, device="cuda", generator=gen)
    d_bits = torch.randint(0, 2, (batch, c.num_inputs), dtype=torch.uint8, device="cuda", generator=gen)
    d_out = torch.zeros((batch, c.num_outputs), dtype=torch.uint8, device="cuda")
    d_mis = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g_ms, e_ms = [], []
    for it in range(4):
        gb.garble(key, d_rnd.data_ptr())
        ev.select_inputs(gb, d_bits.data_ptr())
        ev.eval(key, gb)
        gb.decode(ev, d_out.data_ptr(), d_mis.data_ptr())
        ctx.sync()
        if it:
            g_ms.append(gb.last_ms)
            e_ms.append(ev.last_ms)
    ok = int(d_mis.cpu()[0]) == 0
    bits, out = d_bits.cpu().numpy(), d_out.cpu().numpy()
    for i in (0, batch // 2, batch - 1):
        plain = c.compute_bits(bits[i])  # plaintext evaluation (circuit/computer.go)
        ok = ok and bool((plain[c.NumWires - c.num_outputs:] == out[i]).all())
    g, e = float(np.mean(g_ms)), float(np.mean(e_ms))
    nonfree = info.n_and + info.n_or + info.n_inv
    print(json.dumps({
        "circuit": c.name, "gates": int(info.ngates), "and": int(info.n_and), "levels": int(info.nlevels),
        "hash_phases": int(info.n_hash_phases), "batch": batch, "tile_instances": gb.tile_instances,
        "wires_in_lds": bool(gb.lds_wires), "live_labels": int(info.n_flat_slots) if info.n_flat_slots != 0xffffffff else None,
        "garble_ms": round(g, 4), "eval_ms": round(e, 4),
        "and_gates_per_s": info.n_and * batch / ((g + e) * 1e-3),
        "nonfree_gates_per_s": nonfree * batch / ((g + e) * 1e-3),
        "gates_per_s": info.ngates * batch / ((g + e) * 1e-3),
        "outputs_ok": ok}), flush=True)
    gb.close(); ev.close(); dc.close()
