#!/usr/bin/env python3
"""PCIe-inclusive rate of the literal drop-in calls (gc_garble / gc_eval with HOST buffers): every call moves
the caller's random stream in and R, the input/output wires and the table slab out (garble), the slab and input labels in
and the output labels out (eval).  Reported next to the HBM-resident rate of bench.py, never instead of it."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mpc_amd import engine, parse_file
from mpc_amd.circuit import LABEL, WIRE


def run(batch=1024, reps=5, key=bytes(range(32))):
    c = parse_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "aes_128.gcf"))
    ctx = engine.Context(0)
    dc = engine.DeviceCircuit(ctx, c)
    L = engine.lib()
    rows, nin, nout = dc.info.slab_rows, c.num_inputs, c.num_outputs
    rnd = np.frombuffer(np.random.default_rng(1).bytes(batch * 16 * (nin + 1)), np.uint8).copy()
    k = np.frombuffer(key, np.uint8).copy()
    R = np.zeros(batch, LABEL)
    slab = np.zeros((batch, rows), LABEL)
    io = np.zeros((batch, nin + nout), WIRE)
    outl = np.zeros((batch, nout), LABEL)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    tg, te = [], []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        rc = L.gc_garble(dc.h, p(k), len(k), p(rnd), len(rnd), batch, p(R), None, p(io), p(slab))
        t1 = time.perf_counter()
        assert rc == 0
        inputs = np.ascontiguousarray(io[:, :nin]["l0"])  # all-zero inputs: the L0 labels
        t2 = time.perf_counter()
        rc = L.gc_eval(dc.h, p(k), len(k), batch, None, p(inputs), p(slab), rows, p(outl))
        t3 = time.perf_counter()
        assert rc == 0
        if r:  # the first round warms allocations
            tg.append(t1 - t0)
            te.append(t3 - t2)
    bits = c.compute_bits(np.zeros(nin, np.uint8))[c.NumWires - nout:].astype(bool)  # plaintext result for zero inputs
    want = np.where(bits[None, :], io[:, nin:]["l1"], io[:, nin:]["l0"])
    assert (outl == want).all()
    g, e = min(tg), min(te)
    n_and = dc.info.n_and * batch
    res = {"batch": batch, "garble_ms": g * 1e3, "eval_ms": e * 1e3, "and_gates_per_s": n_and / (g + e),
           "slab_MB": slab.nbytes / 1e6, "garble_GBs_out": (slab.nbytes + io.nbytes) / g / 1e9,
           "eval_GBs_in": (slab.nbytes + inputs.nbytes) / e / 1e9}
    dc.close()
    ctx.close()
    return res


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)))
