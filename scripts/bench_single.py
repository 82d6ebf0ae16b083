#!/usr/bin/env python3
"""Latency of the literal single-instance calls (gc_garble / gc_eval with host buffers, batch = 1): what an unmodified
caller of circuit.Garble / circuit.Eval sees per call."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mpc_amd import engine, parse_file
from mpc_amd.circuit import LABEL, WIRE

p = lambda a: a.ctypes.data_as(C.c_void_p)


def run(circuit="aes_128.gcf", reps=200, key=bytes(range(32))):
    c = parse_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", circuit))
    L = engine.lib()
    ctx = engine.Context(0)
    dc = engine.DeviceCircuit(ctx, c)
    rows, nin, nout = dc.info.slab_rows, c.num_inputs, c.num_outputs
    rnd = np.frombuffer(np.random.default_rng(1).bytes(16 * (nin + 1)), np.uint8).copy()
    k = np.frombuffer(key, np.uint8).copy()
    R = np.zeros(1, LABEL); slab = np.zeros((1, rows), LABEL); io = np.zeros((1, nin + nout), WIRE)
    inputs = np.zeros((1, nin), LABEL); outl = np.zeros((1, nout), LABEL)
    tg, te = [], []
    for r in range(reps + 5):
        t0 = time.perf_counter()
        rc = L.gc_garble(dc.h, p(k), len(k), p(rnd), len(rnd), 1, p(R), None, p(io), p(slab))
        t1 = time.perf_counter()
        assert rc == 0
        inputs[...] = io[:, :nin]["l0"]
        t2 = time.perf_counter()
        rc = L.gc_eval(dc.h, p(k), len(k), 1, None, p(inputs), p(slab), rows, p(outl))
        t3 = time.perf_counter()
        assert rc == 0
        if r >= 5:
            tg.append(t1 - t0); te.append(t3 - t2)
    bits = c.compute_bits(np.zeros(nin, np.uint8))[c.NumWires - nout:].astype(bool)
    want = np.where(bits[None, :], io[:, nin:]["l1"], io[:, nin:]["l0"])
    assert (outl == want).all()
    res = {"circuit": circuit, "and": int(dc.info.n_and), "garble_us_median": float(np.median(tg)) * 1e6,
           "eval_us_median": float(np.median(te)) * 1e6, "garble_us_min": min(tg) * 1e6, "eval_us_min": min(te) * 1e6}
    dc.close(); ctx.close()
    return res


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["aes_128.gcf", "add64.gcf", "sha256xor.gcf"]):
        print(json.dumps(run(name)), flush=True)
