"""Chain fusion on the GPU (mpc_amd/csrc/stream_fuse.cpp): queued steps that depend on ONE earlier queued step are appended
to its launch unit and the chain runs as one planned job.  Nothing the reference's serial loop produces may change
(circuit/stream_garble.go:161-192, 195-449; circuit/stream_evaluator.go:226-432): every byte in program order, every wire
afterwards, the evaluator's labels — compared with the oracle's restatement, across read-after-write / write-after-write /
write-after-read inside a chain and across its boundary, chains cut by the window, and chains whose merged plan fits no
workgroup."""
import os

import numpy as np
import pytest

import oracle
from mpc_amd import engine
from mpc_amd.circuit import adder, bitwise, multiplier, subtractor, synthetic_levelised
from tests.util import drbg

pytestmark = pytest.mark.gpu


def _chain_program(base, bits=32):
    """sums of products with a carry chain behind them (the shape of ed25519.mpcl's FeMul / FeCombine, :270-439, at `bits`
    bits), values overwritten while earlier queued steps still read them, a value written twice inside one chain"""
    mul, add, sub = multiplier(bits), adder(bits), subtractor(bits)
    andb, xorb = bitwise(bits, 2), bitwise(bits, 0)
    nxt = [base + 2]
    prim = [base, base + 1]  # a zero and a one wire for constants

    def fresh(n=bits):
        w = list(range(nxt[0], nxt[0] + n))
        nxt[0] += n
        return w

    def const(v):
        return [prim[1] if (v >> i) & 1 else prim[0] for i in range(bits)]

    vals = []
    for _ in range(12):
        v = fresh()
        prim.extend(v)
        vals.append(v)
    steps = []

    def op(c, a, b, out=None):
        out = fresh(c.num_outputs) if out is None else out
        steps.append((c, a + b, out))
        return out

    def sar(x, k):
        return x[k:] + [x[-1]] * k

    def shl(x, k):
        return [prim[0]] * k + x[:len(x) - k]

    # 1. four sums of four products each: the products are independent, every sum is a chain of three adders
    h = []
    for k in range(4):
        acc = None
        for i in range(4):
            p = op(mul, vals[(k + i) % 6], vals[6 + (k * i) % 6])
            acc = p if acc is None else op(add, acc, p)
        h.append(acc)
    # 2. a carry chain through them (c = (h + 2^(s-1)) >> s; h' += c; h -= c << s), the last carry times 19
    for i in range(3):
        c = sar(op(add, h[i], const(1 << 7)), 8)
        h[i + 1] = op(add, h[i + 1], c)
        h[i] = op(sub, h[i], shl(c, 8))
    c = sar(op(add, h[3], const(1 << 6)), 7)
    h[0] = op(add, h[0], op(mul, c, const(19)))
    h[3] = op(sub, h[3], shl(c, 7))
    # 3. write-after-write INSIDE a chain: t is written, read, written again by a later link, read again
    t = fresh()
    op(add, h[0], h[1], t)
    u = op(xorb, t, vals[0])
    op(add, u, vals[1], t)           # t again: the first store of t must not survive
    v = op(andb, t, u)
    # 4. write-after-read across the chain's boundary: a link overwrites wires that its own chain's head read from the store
    w = op(add, vals[2], vals[3])
    x = op(sub, w, vals[4], vals[2])  # overwrites vals[2], which `w`'s step (same chain) read
    y = op(add, x, vals[2])           # reads the new vals[2] (= x)
    # 5. a conditional-move chain (ed25519.mpcl:69-81: XOR / AND / XOR per link), eight links, one-phase steps
    m = op(sub, const(0), [v[0]] + [prim[0]] * (bits - 1))
    tt = y
    for i in range(8):
        tt = op(xorb, tt, op(andb, op(xorb, tt, vals[5 + i % 6]), m))
    # 6. a value two different chains feed: both must be done before the reader (not fused: two units of one group)
    a1 = op(add, h[2], vals[0])
    a2 = op(add, h[3], vals[1])
    op(add, a1, a2)
    return steps, prim


def _run(ctx, steps, prim, key, rnd, window, by_handle=True, waits=None):
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    first = {w: gg.get(w)["l0"] for w in prim}
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    got, issued, handles = [], 0, {}
    for k in range(len(steps)):
        while issued < min(len(steps), k + window):
            c, in_, out_ = steps[issued]
            if by_handle:
                if id(c) not in handles:
                    handles[id(c)] = gg.intern(c.Gates, c.NumWires, len(in_), len(out_))
                gg.garble_begin_h(handles[id(c)], in_, out_)
            else:
                gg.garble_begin(c.Gates, c.NumWires, in_, out_)
            issued += 1
        got.append(gg.garble_finish())
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, "stream bytes of step %d of %d (%s), window %d" % (k, len(steps), steps[k][0].name, window)
    wires = sorted({o for _, _, out_ in steps for o in out_} | set(prim))
    for o in wires:
        assert gg.get(o) == og.get(o), "garbler's wire %d" % o
    # the evaluator over the same blocks, everything handed over before the first read-back
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    for w in prim:
        ge.set(w, first[w])
        oe.set(w, first[w])
    for (c, in_, out_), b in zip(steps, got):
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, b) == len(b)
        assert oe.circuit(c.NumGates, c.NumWires, nw, b) == len(b)
    for o in wires:
        assert ge.get(o) == oe.get(o), "evaluated label of wire %d" % o
    st = (gg.stats(), gg.fuse_stats(), ge.fuse_stats())
    if waits is not None:
        waits.append((gg.wait_stats(), ge.wait_stats()))
    gg.close()
    ge.close()
    return st


@pytest.mark.parametrize("base,keylen,window,by_handle", [(0, 32, 300, True), (0x20000, 16, 300, False), (0xfff0, 24, 9, True),
                                                          (0, 32, 2, True), (0x20000, 32, 1, False)])
def test_fused_chains_match_oracle(base, keylen, window, by_handle):
    ctx = engine.Context(0)
    steps, prim = _chain_program(base)
    key = drbg("fkey", keylen)
    rnd = drbg("fuse%d" % base, 16 * (len(prim) + 1))
    (groups, grouped, bigs), gf, ef = _run(ctx, steps, prim, key, rnd, window, by_handle)
    assert grouped == len(steps) and bigs == 0
    if window >= 300 and not os.environ.get("GC_STREAM_NO_FUSE"):
        # the chains really ran fused — on both sides — and every one had a one-workgroup plan
        assert gf[0] >= 8 and gf[1] >= 40 and gf[3] == 0, gf
        assert ef[0] >= 8 and ef[1] >= 40 and ef[3] == 0, ef
        assert groups <= 12, groups  # (step by step the carry chain alone is a dozen groups)
    if window == 1:
        assert gf[0] == 0  # nothing queued ahead: nothing to append to
    # later streams of the same ctx find the merged plans (cached per ctx; a chain is planned when it is met the second time,
    # so the second stream may still plan the shapes the first one met once)
    if window >= 300 and not os.environ.get("GC_STREAM_NO_FUSE"):
        assert gf[2] >= 1 and ef[2] >= 1, (gf, ef)  # the four sums have one shape: planned inside the first stream
        _run(ctx, steps, prim, key, rnd, window, by_handle)
        _, gf3, ef3 = _run(ctx, steps, prim, key, rnd, window, by_handle)
        assert gf3[0] == gf[0] and gf3[2] == 0 and ef3[2] == 0, (gf3, ef3)
    ctx.close()


def _dag_program(base, bits=32):
    """steps that conflict with SEVERAL queued steps at once (no chain to be fused into): diamonds, a multiplier behind two
    chains, a wide block behind ten units, a value three units read and a fourth overwrites, the same wires written by three
    steps in a row — and chains that go on behind such steps"""
    mul, add, sub = multiplier(bits), adder(bits), subtractor(bits)
    xorb = bitwise(bits, 0)
    wide = synthetic_levelised(3, 64, 0.4, seed=91, ninputs=bits * 10, inv_frac=0.05)
    nxt = [base + 2]
    prim = [base, base + 1]

    def fresh(n=bits):
        w = list(range(nxt[0], nxt[0] + n))
        nxt[0] += n
        return w

    vals = []
    for _ in range(12):
        v = fresh()
        prim.extend(v)
        vals.append(v)
    steps = []

    def op(c, a, b, out=None):
        out = fresh(c.num_outputs) if out is None else out
        steps.append((c, a + b, out))
        return out

    for rnd in range(3):
        a = [op(add, vals[i], vals[i + 1]) for i in range(10)]          # ten independent units
        b = [op(sub, a[i], a[i + 1]) for i in range(9)]                 # each behind two of them
        c = op(mul, b[0], b[1])                                         # a large step behind two units
        d = op(add, op(add, c[:bits], vals[0]), vals[1])                # a chain that goes on behind it
        steps.append((wide, sum(a, []), fresh(wide.num_outputs)))       # behind ten units: more than a step may name
        w = steps[-1][2]
        r = [op(xorb, vals[2], b[i]) for i in range(3)]                 # three units read vals[2] ...
        op(add, d, w[:bits], vals[2])                                   # ... and this one overwrites it
        t = fresh()
        op(add, r[0], r[1], t)                                          # the same wires written three times in a row,
        op(sub, r[1], r[2], t)                                          # by steps that do not read one another
        op(add, vals[2], r[2], t)
        e = op(add, t, vals[3])
        vals[4 + rnd] = op(mul, e, b[8])[:bits]                          # feeds the next round
        vals[0] = op(add, e, a[9], vals[0])                             # in place: vals[0] is read by queued steps
    return steps, prim


@pytest.mark.parametrize("base,keylen,window", [(0, 32, 400), (0x30000, 16, 400), (0, 24, 7), (0x30000, 32, 1)])
def test_units_that_wait_inside_a_launch_match_oracle(base, keylen, window, monkeypatch):
    """GC_STREAM_DEPS=1 (an experiment, off by default; gc_stream_wait_stats): steps that conflict with several units of an open
    group join it and wait on the device; bytes, wires and evaluated labels are the serial loop's (stream_garble.go:131-157,
    161-192), whatever the window cuts"""
    monkeypatch.setenv("GC_STREAM_DEPS", "1")
    ctx = engine.Context(0)
    steps, prim = _dag_program(base)
    key = drbg("wkey", keylen)
    rnd = drbg("wait%d" % base, 16 * (len(prim) + 1))
    waits = []
    for rep in range(2):
        (groups, grouped, bigs), gf, ef = _run(ctx, steps, prim, key, rnd, window, waits=waits)
        assert grouped == len(steps) and bigs == 0
    if window >= 400 and not os.environ.get("GC_STREAM_NO_FUSE"):
        assert waits[0][0] >= 30 and waits[0][1] >= 30, waits
        assert groups <= 6, groups  # (level by level the program is some forty groups deep)
    if window == 1:
        assert waits[0][0] == 0
    ctx.close()


def test_chain_without_a_one_workgroup_plan_runs_step_by_step():
    """a chain whose merged plan does not fit a workgroup's LDS (wide random circuits: thousands of live labels each) is
    remembered as such and runs one launch per link — with the same bytes"""
    ctx = engine.Context(0)
    shapes = [synthetic_levelised(5, 200, 0.5, seed=70 + v, ninputs=200, inv_frac=0.05) for v in range(3)]
    assert all(c.NumGates <= 1024 for c in shapes)
    prim = list(range(200))
    nxt = [0x300]
    steps, prev = [], prim
    for k in range(14):  # every step reads the 200 outputs of the one before: one chain, 14 x ~1 000 live labels
        c = shapes[k % 3]
        out = list(range(nxt[0], nxt[0] + c.num_outputs))
        nxt[0] += c.num_outputs
        steps.append((c, (prev + prim)[:200], out))
        prev = out
    key, rnd = drbg("unfit", 32), drbg("unfit-r", 16 * (len(prim) + 1))
    (groups, grouped, bigs), gf, ef = _run(ctx, steps, prim, key, rnd, 64)
    assert grouped == len(steps)
    if not os.environ.get("GC_STREAM_NO_FUSE"):
        assert gf[0] >= 1 and ef[0] >= 1  # (whether a unit was unfit depends on the caps of the fusion: either way the bytes hold)
    ctx.close()


def test_fused_chain_cut_by_a_read_back_and_by_set_wire():
    """a read-back (GetInput) launches what is queued: the chain is cut there and goes on in a new unit"""
    ctx = engine.Context(0)
    add = adder(32)
    prim = list(range(2, 2 + 32 * 6))
    vals = [prim[32 * i:32 * i + 32] for i in range(6)]
    key, rnd = drbg("cut", 16), drbg("cut-r", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    nxt, acc, steps = 0x10000, vals[0], []
    for i in range(1, 6):
        out = list(range(nxt, nxt + 32))
        nxt += 32
        steps.append((add, acc + vals[i], out))
        acc = out
    want = [og.garble(c.Gates, c.NumWires, i, o) for c, i, o in steps]
    for k, (c, i, o) in enumerate(steps):
        gg.garble_begin(c.Gates, c.NumWires, i, o)
        if k == 2:
            assert gg.get(o[5]) == og.get(o[5])  # in the middle of the chain
    got = [gg.garble_finish() for _ in steps]
    assert got == want
    for _, _, o in steps:
        for w in o:
            assert gg.get(w) == og.get(w)
    gg.close()
    ctx.close()


def test_default_planning_in_the_background_keeps_the_bytes(monkeypatch):
    """the product's default: a chain runs its steps one launch after the other until the ctx's planner thread has its merged
    plan (asked for at the second meeting); whichever way a unit runs, the bytes and labels are the oracle's"""
    import time
    monkeypatch.delenv("GC_STREAM_FUSE_EAGER", raising=False)
    ctx = engine.Context(0)
    steps, prim = _chain_program(0x4000)
    key, rnd = drbg("bg", 32), drbg("bg-r", 16 * (len(prim) + 1))
    asked = 0
    for rep in range(5):
        _, gf, ef = _run(ctx, steps, prim, key, rnd, 300)
        asked += gf[2] + ef[2]
        time.sleep(0.3)  # (the planner catches up between the streams)
    assert asked >= 2, asked
    _, gf, ef = _run(ctx, steps, prim, key, rnd, 300)
    assert gf[2] == 0 and ef[2] == 0  # everything that repeats is planned by now
    ctx.close()


def _framed_stream(steps, blocks):
    import struct
    return b"".join(struct.pack(">5I", 1, k, c.NumGates, c.NumWires, max(max(in_), max(out_)) + 1) + bytes(b)
                    for k, ((c, in_, out_), b) in enumerate(zip(steps, blocks))) + struct.pack(">I", 2)


@pytest.mark.parametrize("pinned,piece", [(True, 1 << 20), (False, 1 << 20), (True, 150_000), (False, 70_000)])
def test_read_buffers_matched_on_the_device(pinned, piece):
    """gc_stream_eval_blocks on read buffers of many blocks: the DEVICE recognises the blocks (stream_eval_dev.cpp), the host
    reads headers and ids, the rows are gathered from the device copy of the stream — pinned buffers by DMA in place, pageable
    ones through staging; pieces that cut blocks in two; the labels are the oracle's, whichever path a block takes"""
    ctx = engine.Context(0)
    steps, prim = _chain_program(0x8000)
    steps = steps * 4  # (the same instructions again: wires are overwritten, every block is known from the second round on)
    key, rnd = drbg("devm", 32), drbg("devm-r", 16 * (len(prim) + 1))
    og = oracle.Stream(key, rnd, prim)
    first = {w: og.get(w)["l0"] for w in prim}
    blocks = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    framed = _framed_stream(steps, blocks)
    oe = oracle.StreamEval(key)
    for w in prim:
        oe.set(w, first[w])
    for (c, in_, out_), b in zip(steps, blocks):
        assert oe.circuit(c.NumGates, c.NumWires, max(max(in_), max(out_)) + 1, b) == len(b)
    wires = sorted({o for _, _, out_ in steps for o in out_} | set(prim))
    hold = engine.PinnedArray((len(framed),), np.uint8) if pinned else None
    buf = hold.a if pinned else np.frombuffer(framed, np.uint8).copy()
    buf[:] = np.frombuffer(framed, np.uint8)
    for rep in range(2):  # (the second evaluator of the ctx finds the merged plans of the chains)
        ge = engine.StreamEval(ctx, key)
        for w in prim:
            ge.set(w, first[w])
        at, done, win = 0, 0, piece
        while done < len(steps):
            used, nb, more = ge.blocks_at(buf.ctypes.data + at, min(win, len(framed) - at))
            assert used or more, (at, win)
            win = piece if used else win * 2
            at, done = at + used, done + nb
        assert at == len(framed) - 4
        for o in wires:
            assert ge.get(o) == oe.get(o), "label of wire %d" % o
        dev_blocks, fallbacks = ge.dev_stats()
        parsed, matched = ge.stats()
        if not os.environ.get("GC_STREAM_NO_DEVICE_MATCH") and piece >= 150_000:  # (a buffer of fewer than four blocks stays on the host)
            assert dev_blocks >= len(steps) // 2, (dev_blocks, fallbacks, parsed, matched)
        ge.close()
    if hold is not None:
        hold.close()
    ctx.close()


def test_deferred_copies_hand_out_the_same_bytes():
    """gc_stream_garble_finish_async: the copies into the caller's buffer run on the stream's copier threads; after
    gc_stream_garble_copies_wait every byte is the oracle's — mixed with copying finishes and views, slots re-used under way"""
    ctx = engine.Context(0)
    steps, prim = _chain_program(0x2000)
    steps = steps * 3
    key, rnd = drbg("defer", 32), drbg("defer-r", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    want = [bytes(og.garble(c.Gates, c.NumWires, in_, out_)) for c, in_, out_ in steps]
    total = sum(len(w) for w in want)
    out = np.zeros(total + 64, np.uint8)
    handles, issued, off, spans, direct = {}, 0, 0, [], {}
    for k in range(len(steps)):
        while issued < min(len(steps), k + 40):
            c, in_, out_ = steps[issued]
            if id(c) not in handles:
                handles[id(c)] = gg.intern(c.Gates, c.NumWires, len(in_), len(out_))
            gg.garble_begin_h(handles[id(c)], in_, out_)
            issued += 1
        if k % 11 == 5:
            direct[k] = gg.garble_finish()
        elif k % 13 == 7:
            direct[k] = gg.garble_finish_view()
        else:
            n = gg.garble_finish_async(out, off)
            spans.append((k, off, n))
            off += n
        if k % 50 == 49:
            gg.copies_wait()
    gg.copies_wait()
    for k, o, n in spans:
        assert out[o:o + n].tobytes() == want[k], "step %d (deferred copy)" % k
    for k, b in direct.items():
        assert b == want[k], "step %d" % k
    gg.close()
    ctx.close()
