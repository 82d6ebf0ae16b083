"""One random QUEUED streaming program across the scheduling classes (tests/ext_fuzz.py, seventh pass; tests/test_gpu_fuzz.py runs
a slice): step groups on the ctx stream, deep steps on their lanes (threshold lowered so that random circuits of a few dozen
dependent phases count as deep), now and then a big step; results overwrite live wires, operands repeat, some steps update their
inputs in place; 1 to 300 steps queued ahead, 0 to 3 lanes, by handle or by content.  run_case(seed, **overrides) compares every
byte in program order, every wire afterwards and the evaluator's labels with the oracle's serial loop; overrides (window, lanes,
deep_steps, by_handle) replay a seed under other scheduling."""
import os

import numpy as np

import oracle
from mpc_amd import engine
from mpc_amd.circuit import synthetic_levelised
from tests.test_gpu_fuzz import random_circuit
from tests.util import drbg


def make_case(seed):
    rng = np.random.default_rng(123000 + seed)
    cfg = {"deep_steps": int(rng.choice([4, 12, 40])), "lanes": str(rng.choice(["", "0", "1", "2", "3"])),
           "keylen": int(rng.choice([16, 24, 32]))}
    shapes = [random_circuit(rng, int(rng.integers(2, 40)), int(rng.integers(1, 2500)), p_xor=float(rng.choice([0.3, 0.7, 0.9])),
                             reuse=0.0, nout=int(rng.integers(1, 24))) for _ in range(int(rng.integers(2, 7)))]
    if rng.random() < 0.3:
        shapes.append(synthetic_levelised(18, 2048, 0.25, seed=int(rng.integers(1, 1 << 30)), ninputs=64, inv_frac=0.05))
    base = int(rng.choice([0, 0xff00, 0x10000]))
    npool = int(rng.integers(40, 400))
    prim = [base + i for i in range(npool)]
    pool = list(prim)
    nextid = base + npool
    steps, kinds = [], []
    for k in range(int(rng.integers(5, 120))):
        c = shapes[int(rng.integers(0, len(shapes)))]
        in_ = [int(pool[int(rng.integers(0, len(pool)))]) for _ in range(c.num_inputs)]
        u = rng.random()
        if u < 0.45:      # fresh wires
            out_ = list(range(nextid, nextid + c.num_outputs)); nextid += c.num_outputs
            kinds.append("fresh")
        elif u < 0.9 and len(pool) >= c.num_outputs:     # overwrite live wires (distinct ones)
            out_ = [int(x) for x in rng.choice(pool, c.num_outputs, replace=False)]
            kinds.append("overwrite")
        else:             # in place: the first outputs are wires of in[]
            uniq = list(dict.fromkeys(in_))[: c.num_outputs]
            out_ = uniq + list(range(nextid, nextid + c.num_outputs - len(uniq))); nextid += c.num_outputs - len(uniq)
            kinds.append("in place")
        # only outputs a gate really writes join the pool: an output wire that is an input wire of the circuit is never Set, and
        # a never-set global wire reads as (L0, L1) = (0, 0) in the reference but as (0, R) here (L1 is always L0 ^ R)
        for j, o in enumerate(out_):
            if o not in pool and c.NumWires - c.num_outputs + j >= c.num_inputs:
                pool.append(o)
        steps.append((c, in_, out_))
    cfg["window"] = int(rng.choice([1, 2, 7, 33, 300]))
    cfg["by_handle"] = bool(rng.random() < 0.5)
    return cfg, steps, prim, pool, kinds


def run_case(seed, **over):
    cfg, steps, prim, pool, kinds = make_case(seed)
    cfg.update(over)
    os.environ["GC_STREAM_DEEP_STEPS"] = str(cfg["deep_steps"])
    if cfg["lanes"]:
        os.environ["GC_STREAM_DEEP_LANES"] = str(cfg["lanes"])
    else:
        os.environ.pop("GC_STREAM_DEEP_LANES", None)
    what = "steps %d window %d lanes %s deep >= %d %s" % (len(steps), cfg["window"], cfg["lanes"] or "default", cfg["deep_steps"],
                                                        "handle" if cfg["by_handle"] else "content")
    cx = engine.Context(0)  # (the lanes belong to the context: a fresh one per case)
    try:
        key = drbg("qk%d" % seed, cfg["keylen"])
        rnd = drbg("qr%d" % seed, 16 * (len(prim) + 1))
        og, gg = oracle.Stream(key, rnd, prim), engine.Stream(cx, key, rnd, prim)
        oe, ge = oracle.StreamEval(key), engine.StreamEval(cx, key)
        first = {}  # (the labels the program starts from: later steps overwrite live wires)
        for w in prim:
            l = first[w] = gg.get(w)["l0"]
            ge.set(w, l); oe.set(w, l)
        want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
        got, issued, handles = [], 0, {}
        for k in range(len(steps)):
            while issued < min(len(steps), k + cfg["window"]):
                c, in_, out_ = steps[issued]
                if cfg["by_handle"]:
                    if id(c) not in handles:
                        handles[id(c)] = gg.intern(c.Gates, c.NumWires, len(in_), len(out_))
                    gg.garble_begin_h(handles[id(c)], in_, out_)
                else:
                    gg.garble_begin(c.Gates, c.NumWires, in_, out_)
                issued += 1
            got.append(gg.garble_finish())
        for k, (g, w) in enumerate(zip(got, want)):
            assert g == w, "%s: stream bytes of step %d (%s, %d gates, circuit %s)" % (what, k, kinds[k], steps[k][0].NumGates, steps[k][0].name)
        for o in pool[::3]:
            assert gg.get(o) == og.get(o), "%s: garbler's wire %d" % (what, o)
        for (c, in_, out_), b in zip(steps, got):
            nw = max(max(in_), max(out_)) + 1
            assert ge.circuit(c.NumGates, c.NumWires, nw, b) == len(b)
            assert oe.circuit(c.NumGates, c.NumWires, nw, b) == len(b)
        for o in pool[::3]:
            assert ge.get(o) == oe.get(o), "%s: evaluated label of wire %d" % (what, o)
        # the same stream as the peer frames it, through gc_stream_eval_blocks in pieces of a size drawn per case (from "most
        # blocks are cut" to "all in one call"): the same labels
        import struct
        import numpy as np
        gb = engine.StreamEval(cx, key)
        for w in prim:
            gb.set(w, first[w])
        framed = b"".join(struct.pack(">5I", 1, k, c.NumGates, c.NumWires, max(max(in_), max(out_)) + 1) + bytes(b)
                          for k, ((c, in_, out_), b) in enumerate(zip(steps, got))) + struct.pack(">I", 2)
        rng = np.random.default_rng(seed)
        piece = int(rng.choice([700, 5000, 70000, 1 << 20, len(framed)]))
        pos, done, win = 0, 0, piece
        while done < len(steps):
            used, nb, more = gb.blocks(framed[pos:pos + win])
            assert used or more, "%s: gc_stream_eval_blocks stopped at byte %d of %d without asking for more" % (what, pos, len(framed))
            win = piece if used else win * 2
            pos, done = pos + used, done + nb
        assert pos == len(framed) - 4, "%s: %d bytes of the framed stream consumed, %d expected" % (what, pos, len(framed) - 4)
        for o in pool[::3]:
            assert gb.get(o) == oe.get(o), "%s: label of wire %d after gc_stream_eval_blocks" % (what, o)
        cx.sync()
        gg.close(); ge.close(); gb.close()
    finally:
        cx.close()
        os.environ.pop("GC_STREAM_DEEP_STEPS", None)
        os.environ.pop("GC_STREAM_DEEP_LANES", None)
    return what
