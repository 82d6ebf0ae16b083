#!/usr/bin/env python3
"""What could ANY scheduler reach on a streamed program?  (priced before building a device-side job queue.)

Every step of the program costs what ONE workgroup takes for it today (1.3 us per dependent hash phase, 9.7 ns per
gate where a phase is wider than the workgroup, ~8 us of job set-up; a 131 072-gate step 0.29 ms as a cooperative launch) and
depends on the steps it has a RAW / WAW / WAR relation with through its global wires.  The model then runs the program on a
machine with INFINITELY many workgroups and NO launch cost, under the one constraint the caller imposes: at most `window` steps
begun and not yet finished, bytes handed out in program order (gc_stream_garble_begin / _finish; window 1 = the unchanged
caller).  No GPU.

usage: scripts/stream_ideal_model.py [program ...]     (names of scripts/bench_stream.py: ssa23 mixed ed25519like ...)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import bench_stream as bs  # noqa: E402
from mpc_amd import engine  # noqa: E402
from mpc_amd.circuit import GATE  # noqa: E402

T_PHASE, T_GATE, T_JOB, T_BIG = 1.3e-6, 9.7e-9, 8e-6, 0.29e-3


def model(name):
    steps, _ = bs.PROGRAMS[name]()
    dur = {}

    def duration(c):
        if id(c) not in dur:
            if (c.name or "").startswith("synth_") and c.NumGates >= 100000:  # a wide step without a one-workgroup plan: a cooperative launch
                dur[id(c)] = T_BIG
            else:
                p = engine.Plan(np.ascontiguousarray(c.Gates, dtype=GATE), c.NumWires, c.num_inputs, c.num_outputs)
                dur[id(c)] = max(p.info.n_hash_phases * T_PHASE, c.NumGates * T_GATE) + T_JOB
        return dur[id(c)]
    n = len(steps)
    D = np.array([duration(c) for c, _, _ in steps])
    gates = sum(c.NumGates for c, _, _ in steps)
    last_w, last_r, deps = {}, {}, []
    for k, (c, in_, out_) in enumerate(steps):
        d = set()
        for w in in_:
            if w in last_w:
                d.add(last_w[w])
        for w in out_:
            if w in last_w:
                d.add(last_w[w])
            d.update(last_r.get(w, ()))
        d.discard(k)
        deps.append(d)
        for w in in_:
            last_r.setdefault(w, []).append(k)
        for w in out_:
            last_w[w], last_r[w] = k, []
    print("%s: %d steps, %.3g gates; all steps on ONE workgroup one after the other: %.1f ms" % (name, n, gates, D.sum() * 1e3))
    for window in (1, 64, 256, 1024, 1 << 30):
        end, F = np.zeros(n), np.zeros(n)
        for k in range(n):
            s = max((end[j] for j in deps[k]), default=0.0)
            if k >= window:
                s = max(s, F[k - window])
            end[k] = s + D[k]
            F[k] = max(F[k - 1] if k else 0.0, end[k])
        print("  window %-10s ideal %.1f ms = %.2e gates/s" % ("unbounded" if window > 1 << 20 else window, F[-1] * 1e3, gates / F[-1]))


if __name__ == "__main__":
    for nm in sys.argv[1:] or ["ssa23", "mixed"]:
        model(nm)
