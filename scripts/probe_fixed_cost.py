#!/usr/bin/env python3
"""Where does the fixed ~1 ms around bench.py's timed loop come from?  Times N graph launches of the step
between two device syncs for several N (the intercept of the fit is the fixed part), with and without a kernel kept in
flight across the fence, and the GPU-side span of the same launches from HIP events."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from mpc_amd import engine, parse_file


def main():
    circ = parse_file(os.path.join(ROOT, "tests", "golden", "aes_128.gcf"))
    key = bytes(range(32))
    ctx = engine.Context(0)
    dc = engine.DeviceCircuit(ctx, circ)
    batch = 1024
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    d_rnd = ctx.random_u8((batch, circ.num_inputs + 1, 16), 256, seed=1)
    d_bits = ctx.random_u8((batch, circ.num_inputs), 2, seed=2)
    d_acc = ctx.zeros((1, batch, circ.num_outputs))
    d_mis = ctx.zeros(1, np.int32)

    def step():
        gb.garble(key, d_rnd)
        ev.select_inputs(gb, d_bits)
        ev.eval(key, gb)
        gb.decode(ev, d_acc, d_mis)

    step()
    ctx.sync()
    g = ctx.capture(step)
    out = {}
    for idle_ms in (0, 2, 20):
        rows = []
        for n in (5, 10, 20, 40, 80, 160, 320):
            best = None
            for rep in range(3):
                for _ in range(5):
                    g.launch()
                ctx.sync()
                if idle_ms:
                    time.sleep(idle_ms / 1e3)
                t0 = time.perf_counter()
                for _ in range(n):
                    g.launch()
                t1 = time.perf_counter()
                ctx.sync()
                t2 = time.perf_counter()
                r = (t2 - t0, t1 - t0)
                best = r if best is None or r[0] < best[0] else best
            rows.append((n, best[0] * 1e3, best[1] * 1e3))
        ns = np.array([r[0] for r in rows], float)
        ts = np.array([r[1] for r in rows], float)
        slope, icpt = np.polyfit(ns, ts, 1)
        out["idle_%dms" % idle_ms] = {"rows_n_totalms_enqueuems": rows, "ms_per_step": slope, "fixed_ms": icpt}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
