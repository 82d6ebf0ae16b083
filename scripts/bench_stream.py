#!/usr/bin/env python3
"""Config-5-shaped streaming garbler throughput (circuit/stream_garble.go path): a program of large per-step circuits
(each step consumes the previous step's outputs through global wire ids) through gc_stream_garble.  A single serial
instance: one launch sequence per step.  Prints one JSON line: gates/s, AND/s, stream bytes/s and the SHA-256 of the
byte stream (tests/test_gpu_stream.py checks the same construction against the CPU restatement at a smaller size)."""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mpc_amd import engine
from mpc_amd.circuit import synthetic_levelised


def make_steps(nsteps, levels, width, and_frac, nin=256):
    """step k reads global wires [k*nin, (k+1)*nin) and writes [(k+1)*nin, (k+2)*nin)"""
    steps = []
    for k in range(nsteps):
        c = synthetic_levelised(levels, width, and_frac, seed=900 + (k % 4), ninputs=nin, inv_frac=0.05)
        # the generator emits min(width, 128) outputs; chain them (repeated) into the next step's nin inputs
        nout = c.num_outputs
        in_ = [k * nin + i for i in range(nin)]
        out_ = [(k + 1) * nin + i for i in range(nout)]
        steps.append((c, in_, out_))
    return steps


def run(total_gates=10_000_000, levels=64, width=2048, and_frac=0.25, key=bytes(range(32)), ctx=None, evaluate=True):
    per = levels * width
    nsteps = max(1, total_gates // per)
    nin = 256
    steps = make_steps(nsteps, levels, width, and_frac, nin)
    prim = list(range(nin))
    # later steps read wires the previous step did not write (nout < nin): declare those as primary inputs as well
    for k in range(1, nsteps):
        prim += [k * nin + i for i in range(steps[k - 1][0].num_outputs, nin)]
    rnd = hashlib.shake_256(b"stream-bench").digest(16 * (len(prim) + 1))
    own = ctx is None
    if own:
        ctx = engine.Context(0)
    g = engine.Stream(ctx, key, rnd, prim)
    ev = engine.StreamEval(ctx, key) if evaluate else None
    if ev is not None:  # the evaluator's input labels: all-zero inputs
        for w in prim:
            wire = g.get(w)
            ev.set(w, (int(wire["l0"]["d0"]), int(wire["l0"]["d1"])))
    etimes = []
    h = hashlib.sha256()
    nbytes = 0
    # The clock runs around the garbler's calls only (hashing the stream for the parity check is not part of the path).
    # Steps are pipelined the way a driver would: gc_stream_garble_begin(k + 1) before gc_stream_garble_finish(k) — the
    # bytes leave in order, the host's share of a step overlaps the GPU's share of the step before.  The first use of a
    # circuit builds and caches its plan: reported separately from the steady state.
    times = []
    keep = []  # the blocks of the first steps, for the evaluator pass
    neval = min(nsteps, 150) if evaluate else 0
    t_prev = time.perf_counter()
    g.garble_begin(steps[0][0].Gates, steps[0][0].NumWires, steps[0][1], steps[0][2])
    for k in range(nsteps):
        if k + 1 < nsteps:
            c1, in1, out1 = steps[k + 1]
            g.garble_begin(c1.Gates, c1.NumWires, in1, out1)
        data = g.garble_finish()
        now = time.perf_counter()
        times.append(now - t_prev)
        h.update(data)
        nbytes += len(data)
        if k < neval:
            keep.append(data)
        t_prev = time.perf_counter()
    if ev is not None:
        # evaluator alone over the stored blocks: a call returns when the block is parsed and its kernels are enqueued, so
        # the clock stops after the final read-back (which waits for everything)
        marks = []
        for k in range(neval):
            c, in_, out_ = steps[k]
            nw = max(max(in_), max(out_)) + 1
            t0 = time.perf_counter()
            used = ev.circuit(c.NumGates, c.NumWires, nw, keep[k])
            etimes.append(time.perf_counter() - t0)
            assert used == len(keep[k])
            marks.append(time.perf_counter())
        t0 = time.perf_counter()
        outs = steps[neval - 1][2][:8]
        got_all = [ev.get(o) for o in outs]  # the last step's outputs must be valid labels of the garbler's wires
        etimes[-1] += time.perf_counter() - t0
        for o, got in zip(outs, got_all):
            wire = g.get(o)
            assert got in ((int(wire["l0"]["d0"]), int(wire["l0"]["d1"])), (int(wire["l1"]["d0"]), int(wire["l1"]["d1"])))
        ev.close()
    gates = sum(c.NumGates for c, _, _ in steps)
    ands = sum(c.stats()["AND"] for c, _, _ in steps)
    g.close()
    if own:
        ctx.close()
    dt = sum(times)
    ndistinct = min(4, nsteps)
    steady = times[ndistinct:]
    per_step_gates = gates / nsteps
    res = {"steps": nsteps, "gates": gates, "and": ands, "seconds": dt, "gates_per_s": gates / dt,
           "and_per_s": ands / dt, "stream_bytes": nbytes, "stream_MBps": nbytes / dt / 1e6, "sha256": h.hexdigest()}
    if steady:
        sdt = sum(steady)
        res["steady_ms_per_step"] = sdt / len(steady) * 1e3
        res["steady_gates_per_s"] = per_step_gates * len(steady) / sdt
        res["first_use_ms_per_circuit"] = sum(times[:ndistinct]) / ndistinct * 1e3
    if etimes[ndistinct:]:
        sdt = sum(etimes[ndistinct:])
        res["eval_steady_ms_per_step"] = sdt / len(etimes[ndistinct:]) * 1e3
        res["eval_steady_gates_per_s"] = per_step_gates * len(etimes[ndistinct:]) / sdt
    return res


def run_for_line(key=bytes(range(32)), ctx=None):
    """the `stream` object of bench.py's line: a bounded sample of the big-step program (152 chained 131 072-gate steps)"""
    st = run(20_000_000, key=key, ctx=ctx)
    return {k: st[k] for k in ("steps", "gates", "steady_ms_per_step", "steady_gates_per_s", "eval_steady_ms_per_step",
                               "eval_steady_gates_per_s", "first_use_ms_per_circuit", "sha256") if k in st}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000)))
