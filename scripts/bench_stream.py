#!/usr/bin/env python3
"""Config-5-shaped streaming throughput (circuit/stream_garble.go path; compiler/ssa/streamer.go:412-524 garbles one
circuit per SSA instruction).  ONE serial instance; three kinds of programs through gc_stream_garble_begin / _finish
(garbler, a window of steps queued ahead) and gc_stream_eval_circuit (evaluator, over the produced bytes):

  big      chained 131 072-gate steps (64 levels x 2 048): every step its own launch sequence, the engine's best case
  uniform  small steps of one shape (512 or 4 096 gates) in `chains` interleaved dependency chains: step k reads the
           outputs of step k - chains, so `chains` independent steps are always available
  mixed    70 % 64-bit adders, 25 % 64 x 64 multipliers, 5 % 131 072-gate steps; operands are drawn mostly from the
           values of the last few steps (dependency chains as SSA code has them), results go to fresh wires or overwrite
           an old variable (wire re-use)

Prints one JSON line per program: gates/s of both sides, launch sequences used, SHA-256 of the byte stream.  The SHA-256s
of the programs bench.py runs are pinned by the ORACLE (tests/golden/stream_bench_golden.json, written by
tests/golden/make_stream_bench_golden.py in the build container) and checked here."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from mpc_amd import engine
from mpc_amd.circuit import GATE, adder, comparator64, multiplier, synthetic_levelised

GOLDEN = os.path.join(ROOT, "tests", "golden", "stream_bench_golden.json")


# ---- programs: lists of (circuit, in wire ids, out wire ids) + the primary input wires ---------------------------------

def program_big(total_gates=20_000_000, levels=64, width=2048, and_frac=0.25, nin=256):
    """step k reads global wires [k*nin, (k+1)*nin) and writes [(k+1)*nin, (k+1)*nin + nout)"""
    nsteps = max(1, total_gates // (levels * width))
    steps = []
    four = [synthetic_levelised(levels, width, and_frac, seed=900 + s, ninputs=nin, inv_frac=0.05) for s in range(min(4, nsteps))]
    for k in range(nsteps):
        c = four[k % 4]
        steps.append((c, [k * nin + i for i in range(nin)], [(k + 1) * nin + i for i in range(c.num_outputs)]))
    prim = list(range(nin))
    # later steps read wires the previous step did not write (nout < nin): declare those as primary inputs as well
    for k in range(1, nsteps):
        prim += [k * nin + i for i in range(steps[k - 1][0].num_outputs, nin)]
    return steps, prim


def make_steps(nsteps, levels, width, and_frac, nin=256):
    """the big-step construction at a chosen size (tests/test_gpu_stream.py)"""
    steps = []
    for k in range(nsteps):
        c = synthetic_levelised(levels, width, and_frac, seed=900 + (k % 4), ninputs=nin, inv_frac=0.05)
        steps.append((c, [k * nin + i for i in range(nin)], [(k + 1) * nin + i for i in range(c.num_outputs)]))
    return steps


def program_uniform(gates=512, nsteps=4000, chains=8):
    """`chains` interleaved chains of one small shape: step k reads what step k - chains wrote"""
    levels, width = {512: (8, 64), 4096: (16, 256)}[gates]
    nin = 64
    shapes = [synthetic_levelised(levels, width, 0.25, seed=700 + v, ninputs=nin, inv_frac=0.05) for v in range(4)]
    nout = shapes[0].num_outputs  # = min(width, 128) >= nin
    prim = list(range(chains * nin))
    steps = []
    base = 0x11000  # 32-bit wire ids, as a program of this length has
    for k in range(nsteps):
        c = shapes[k % 4]
        src = prim[(k % chains) * nin:(k % chains + 1) * nin] if k < chains else steps[k - chains][2][:nin]
        steps.append((c, list(src), [base + k * nout + i for i in range(nout)]))
    return steps, prim


def program_mixed(nsteps=3000, seed=5, big_every=20):
    rng = np.random.default_rng(seed)
    add, mul = adder(64), multiplier(64)
    bigs = [synthetic_levelised(64, 2048, 0.25, seed=900 + v, ninputs=256, inv_frac=0.05) for v in range(2)]
    nvals = 64
    prim = list(range(64 * nvals))
    vals = [prim[64 * v:64 * (v + 1)] for v in range(nvals)]  # the program's live 64-bit values
    recent = []
    nxt = 0x20000
    steps = []

    def operand():
        if recent and rng.random() < 0.6:  # the value of one of the last eight steps: a dependency chain
            return recent[int(rng.integers(max(0, len(recent) - 8), len(recent)))]
        return vals[int(rng.integers(0, len(vals)))]

    for k in range(nsteps):
        u = rng.random()
        if big_every and k % big_every == big_every - 1:
            c = bigs[(k // big_every) % 2]
            in_ = operand() + operand() + operand() + operand()
            nres = 2
        else:
            c = add if u < 0.7368 else mul  # 70 : 25 among the small steps
            in_ = operand() + operand()
            nres = 1
        if rng.random() < 0.3:  # overwrite old variables (the allocator recycles wires)
            tgt = [int(rng.integers(0, len(vals))) for _ in range(nres)]
            while len(set(tgt)) < nres:
                tgt = [int(rng.integers(0, len(vals))) for _ in range(nres)]
            out_ = sum((vals[t] for t in tgt), [])
            if set(out_) & set(in_):  # an in-place update is legal but rare: keep it out of the common path
                out_ = list(range(nxt, nxt + 64 * nres))
                nxt += 64 * nres
                tgt = None
        else:
            out_ = list(range(nxt, nxt + 64 * nres))
            nxt += 64 * nres
            tgt = None
        steps.append((c, in_, out_))
        for r in range(nres):
            v = out_[64 * r:64 * (r + 1)]
            recent.append(v)
            if tgt is None:
                vals[int(rng.integers(0, len(vals)))] = v  # the new value replaces a dead one in the pool
    return steps, prim


def ssa_circuits():
    """23 distinct circuits, as many as the streamed Ed25519 program of the reference caches (benchmarks.md:698): the
    instruction mix of a compiled MPCL program — adders and multipliers from 8 to 512 / 256 bits, a comparator, bitwise
    operations (all-AND, all-OR, all-XOR, NOT: the last two garble without a table), mux- and shift-like blocks and one
    wide hash-like block.  (name, circuit, weight in the instruction mix)"""
    syn = synthetic_levelised
    out = []
    for bits, wgt in ((8, 3), (16, 4), (32, 10), (64, 14), (128, 2), (256, 2), (512, 1)):
        out.append(("add%d" % bits, adder(bits), wgt))
    for bits, wgt in ((8, 2), (16, 3), (32, 6), (64, 6), (128, 2), (256, 1)):
        out.append(("mul%d" % bits, multiplier(bits), wgt))
    out.append(("cmp64", comparator64(), 5))
    out.append(("and64", syn(1, 64, 1.0, seed=41, ninputs=128), 5))
    out.append(("or64", syn(1, 64, 0.0, seed=42, ninputs=128, or_frac=1.0), 3))
    out.append(("xor64", syn(1, 64, 0.0, seed=43, ninputs=128), 7))
    out.append(("and256", syn(1, 256, 1.0, seed=44, ninputs=512), 3))
    out.append(("xor256", syn(1, 256, 0.0, seed=45, ninputs=512), 4))
    out.append(("not256", syn(1, 256, 0.0, seed=46, ninputs=512, inv_frac=1.0), 3))
    out.append(("mux64", syn(3, 64, 0.34, seed=47, ninputs=128), 6))
    out.append(("shift64", syn(6, 64, 0.0, seed=48, ninputs=128), 4))
    out.append(("hash512", syn(16, 512, 0.25, seed=49, ninputs=512, inv_frac=0.05), 2))
    assert len(out) == 23
    return out


def program_ssa(nsteps=6000, seed=11):
    """a stream of `nsteps` instructions drawn from ssa_circuits(): operands are live values of the instruction's width — 60 %
    the result of one of the last eight instructions of that width (dependency chains), else a variable of the pool — and
    30 % of the results overwrite a variable (the allocator recycles wires)"""
    rng = np.random.default_rng(seed)
    circs = ssa_circuits()
    wsum = float(sum(w for _, _, w in circs))
    probs = [w / wsum for _, _, w in circs]
    widths = sorted({c.num_inputs // 2 for _, c, _ in circs})
    nxt = 0
    prim, pool, recent = [], {}, {}
    for w in widths:  # eight variables per width: the program's primary inputs
        pool[w] = []
        for _ in range(8):
            pool[w].append(list(range(nxt, nxt + w)))
            prim += pool[w][-1]
            nxt += w
        recent[w] = []
    nxt = max(nxt, 0x18000)  # results get ids beyond 16 bits, as in a program of this length
    steps = []
    for _ in range(nsteps):
        _, c, _ = circs[int(rng.choice(len(circs), p=probs))]
        w = c.num_inputs // 2

        def operand():
            r = recent[w]
            if r and rng.random() < 0.6:
                return r[int(rng.integers(max(0, len(r) - 8), len(r)))]
            return pool[w][int(rng.integers(0, len(pool[w])))]

        in_ = operand() + operand()
        nout = c.num_outputs
        out_ = None
        if nout == w and rng.random() < 0.3:  # overwrite a variable that is not an operand of this instruction
            t = int(rng.integers(0, len(pool[w])))
            if not set(pool[w][t]) & set(in_):
                out_ = pool[w][t]
        if out_ is None:
            out_ = list(range(nxt, nxt + nout))
            nxt += nout
            if nout == w:
                pool[w][int(rng.integers(0, len(pool[w])))] = out_
        if nout == w:
            recent[w].append(out_)
        elif nout in recent:  # (a 128-bit result of a wider block is a 128-bit value)
            recent[nout].append(out_)
        steps.append((c, in_, out_))
    return steps, prim


class _Ssa:
    """values of a compiled program as lists of global wire ids (LSB first), one circuit per arithmetic instruction
    (compiler/ssa/streamer.go:412-524); casts, shifts by constants and truncations re-name wires and garble nothing
    (streamer.go:330-391: the Lshift / Rshift / Slice cases fill `out` with ids).  Results take their ids from a ring of
    `ring` wires behind the primary inputs: the allocator of the reference recycles wires the same way
    (compiler/ssa/wire_allocator.go), so a long program writes wires that earlier instructions read and wrote."""

    def __init__(self, ring=1 << 21):
        self.zero, self.one = 0, 1          # prog.zeroWire and a wire holding 1: constants are lists of these two
        self.prim = [0, 1]
        self.base = None
        self.ring = ring
        self.pos = 0
        self.steps = []

    def inputs(self, n):
        assert self.base is None, "primary inputs come first"
        ids = list(range(len(self.prim), len(self.prim) + n))
        self.prim += ids
        return ids

    def _new(self, n):
        if self.base is None:
            self.base = max(len(self.prim), 0x10000)  # a program of this length names its wires with 32 bits
        if self.pos + n > self.ring:
            self.pos = 0
        ids = list(range(self.base + self.pos, self.base + self.pos + n))
        self.pos += n
        return ids

    def op(self, circ, a, b):
        out = self._new(circ.num_outputs)
        self.steps.append((circ, a + b, out))
        return out

    def const(self, v, bits):
        return [self.one if (v >> i) & 1 else self.zero for i in range(bits)]

    def sext(self, x, bits=64):   # int64(x)
        return x + [x[-1]] * (bits - len(x))

    def shl(self, x, k):
        return [self.zero] * k + x[:len(x) - k]

    def sar(self, x, k):
        return x[k:] + [x[-1]] * k


def program_ed25519(iterations=10):
    """The inner loop of Ed25519 signing as the reference streams it (benchmarks.md:677-704: `sign.mpcl`, 8.45e8 gates, 32 %
    AND, 23 cached circuits; 96 % of the time in Garble).  pkg/crypto/ed25519/internal/edwards25519/ed25519.mpcl is the ref10
    code: a field element is [10]int32, FeMul (:388-439) forms 100 int64 products from sign-extended limbs and constant
    multiples (2 f_i, 19 g_i: int32 multiplications), adds ten of them per output limb and runs the carry chain of FeCombine
    (:270-385: c = (h + 2^25) >> 26; h' += c; h -= c << 26 — twelve times, two of them in parallel, one carry times 19);
    GeScalarMultBase (:991-1045) repeats per scalar digit: selectPoint (:966-989: eight constant-time conditional moves of a
    table point — FeCMove :69-81 is XOR / AND / XOR per limb — behind equal() :949-954, then the conditional negation),
    geMixedAdd (:832-845: three FeMul, seven FeAdd / FeSub) and ToExtended (:787-792: four FeMul).  This program is
    `iterations` digits of that loop, instruction by instruction, in the reference's dependency order: per digit 2 624
    instructions, 1.04e7 gates, 29 % AND, from eight circuits (64-bit multiplier, adder, subtractor; 32-bit multiplier, adder,
    subtractor, AND, XOR).  The multiplier is the array form of compiler/circuits/circ_multiplier.go:38 (the reference picks
    Karatsuba above ~21 bits, :18-36), adder and subtractor are its one-AND-per-bit full adder / subtractor
    (circ_adder.go:30-55, circ_subtractor.go:17-31).  Small constants (1 << 25, 19, 2, 0, 1) are lists of the zero / one wire."""
    from mpc_amd.circuit import AND as OP_AND, XOR as OP_XOR, bitwise, subtractor
    mul64, add64, sub64 = multiplier(64), adder(64), subtractor(64)
    mul32, add32, sub32 = multiplier(32), adder(32), subtractor(32)
    and32, xor32 = bitwise(32, OP_AND), bitwise(32, OP_XOR)
    P = _Ssa()
    digits = [P.inputs(32) for _ in range(iterations)]   # e[i], sign-extended to int32 (the scalar's signed digits)
    # base[pos][i] (const.mpcl): 8 points x 3 field elements x 10 limbs per digit.  The reference's compiler folds such constants
    # into the instruction's circuit (streamer.go:468-476: ConstPropagate + Prune before the circuit is cached); here a table
    # limb is a 32-bit word of wires of its own, so that the program keeps to its eight circuits (bound to wires of constant
    # value the same XOR block would reach the evaluator in a different repeat pattern for every table entry)
    tables = [[[[P.inputs(32) for _ in range(10)] for _ in range(3)] for _ in range(8)] for _ in range(iterations)]

    def fe_add(a, b):
        return [P.op(add32, x, y) for x, y in zip(a, b)]

    def fe_sub(a, b):
        return [P.op(sub32, x, y) for x, y in zip(a, b)]

    def fe_combine(h):  # :270-385
        h = list(h)

        def carry(i, shift, nxt, times19=False):
            c = P.sar(P.op(add64, h[i], P.const(1 << (shift - 1), 64)), shift)
            h[nxt] = P.op(add64, h[nxt], P.op(mul64, c, P.const(19, 64)) if times19 else c)
            h[i] = P.op(sub64, h[i], P.shl(c, shift))

        for i, j in ((0, 4), (1, 5), (2, 6), (3, 7), (4, 8)):  # the two chains run side by side
            carry(i, 26 if i % 2 == 0 else 25, i + 1)
            carry(j, 26 if j % 2 == 0 else 25, j + 1)
        carry(9, 25, 0, times19=True)
        carry(0, 26, 1)
        return [x[:32] for x in h]  # int32(h_i)

    def fe_mul(f, g):  # :388-439
        f2 = {i: P.sext(P.op(mul32, f[i], P.const(2, 32))) for i in (1, 3, 5, 7, 9)}
        g19 = {i: P.sext(P.op(mul32, g[i], P.const(19, 32))) for i in range(1, 10)}
        F = [P.sext(x) for x in f]
        G = [P.sext(x) for x in g]
        h = []
        for k in range(10):
            acc = None
            for i in range(10):
                j = (k - i) % 10
                a = f2[i] if (i % 2 == 1 and k % 2 == 0) else F[i]
                b = g19[j] if i + j >= 10 else G[j]
                pr = P.op(mul64, a, b)
                acc = pr if acc is None else P.op(add64, acc, pr)
            h.append(acc)
        return fe_combine(h)

    def fe_cmove(f, g, mask):  # :69-81, b already negated into a mask
        return [P.op(xor32, x, P.op(and32, P.op(xor32, x, y), mask)) for x, y in zip(f, g)]

    def select_point(b, tab):  # :966-989
        bneg = P.op(and32, P.sar(b, 31), P.const(1, 32))
        nmask = P.op(sub32, P.const(0, 32), bneg)
        babs = P.op(sub32, b, P.shl(P.op(and32, nmask, b), 1))
        t = [[P.const(1, 32)] + [P.const(0, 32)] * 9, [P.const(1, 32)] + [P.const(0, 32)] * 9, [P.const(0, 32)] * 10]
        for i in range(8):
            x = P.op(sub32, P.op(xor32, babs, P.const(i + 1, 32)), P.const(1, 32))  # equal(): (b ^ c) - 1 >> 31
            mask = P.op(sub32, P.const(0, 32), P.sar(x, 31)[:1] + [P.zero] * 31)
            t = [fe_cmove(t[k], tab[i][k], mask) for k in range(3)]
        minus = [t[1], t[0], [P.op(sub32, P.const(0, 32), x) for x in t[2]]]  # FeCopy, FeCopy, FeNeg
        return [fe_cmove(t[k], minus[k], nmask) for k in range(3)]

    hX, hY, hZ, hT = ([P.const(v, 32)] + [P.const(0, 32)] * 9 for v in (0, 1, 1, 0))  # h.Zero()
    for it in range(iterations):
        ypx, ymx, xy2d = select_point(digits[it], tables[it])
        # geMixedAdd (:832-845)
        rX = fe_add(hY, hX)
        rY = fe_sub(hY, hX)
        rZ = fe_mul(rX, ypx)
        rY = fe_mul(rY, ymx)
        rT = fe_mul(xy2d, hT)
        t0 = fe_add(hZ, hZ)
        rX = fe_sub(rZ, rY)
        rY = fe_add(rZ, rY)
        rZ = fe_add(t0, rT)
        rT = fe_sub(t0, rT)
        # ToExtended (:787-792)
        hX, hY, hZ, hT = fe_mul(rX, rT), fe_mul(rY, rZ), fe_mul(rZ, rT), fe_mul(rX, rY)
    return P.steps, P.prim


PROGRAMS = {
    "ed25519like": lambda: program_ed25519(10),
    "ed25519like1": lambda: program_ed25519(1),   # one digit: the parity tests' size
    "ssa23": lambda: program_ssa(6000),
    "big": lambda: program_big(20_000_000),
    "big130": lambda: program_big(130_000_000),
    "uniform512": lambda: program_uniform(512, 4000, 8),
    "uniform4096": lambda: program_uniform(4096, 1500, 8),
    "uniform512x64": lambda: program_uniform(512, 8000, 64),
    "mixed": lambda: program_mixed(3000),
    "mixed100": lambda: program_mixed(10000),  # config 5's size: adders / multipliers / wide steps, >= 1e8 gates in all
}


# what a program looks like (the driver line says which rows are friendly synthetic shapes and which are
# instruction mixes) and how far ahead its driver queues (steps begun and not finished; the Ed25519 program needs the hundred
# products of a field multiplication in flight together)
SHAPES = {"big": "wide-synthetic", "big130": "wide-synthetic", "uniform512": "uniform-synthetic", "uniform4096": "uniform-synthetic",
          "uniform512x64": "uniform-synthetic", "mixed": "instruction-mix", "mixed100": "instruction-mix", "ssa23": "instruction-mix",
          "ed25519like": "instruction-mix (ref10 Ed25519 inner loop)", "ed25519like1": "instruction-mix (ref10 Ed25519 inner loop)"}
WINDOWS = {"ed25519like": 1024}


def stream_rnd(name, nprim):
    return hashlib.shake_256(b"stream-bench/" + name.encode()).digest(16 * (nprim + 1))


# ---- lean drivers: one ctypes call per step, arguments converted once -------------------------------------------------

class _Args:
    def __init__(self, steps):
        self.keep = []
        self.begin = []
        circs = {}
        for c, in_, out_ in steps:
            if id(c) not in circs:
                g = np.ascontiguousarray(c.Gates, dtype=GATE)
                circs[id(c)] = (g, g.ctypes.data_as(C.c_void_p))
            g, pg = circs[id(c)]
            i = np.asarray(in_, np.uint32)
            o = np.asarray(out_, np.uint32)
            self.keep.append((g, i, o, c))
            self.begin.append((pg, len(g), c.NumWires, i.ctypes.data_as(C.c_void_p), len(i), o.ctypes.data_as(C.c_void_p), len(o)))

    def interned(self, stream):
        """the same steps as (handle, in, out) for gc_stream_garble_begin_h: what a Go host with a map[*Circuit]handle passes"""
        handles, out = {}, []
        for (g, i, o, c), b in zip(self.keep, self.begin):
            if id(c) not in handles:
                handles[id(c)] = stream.intern(g, c.NumWires, len(i), len(o))
            out.append((handles[id(c)], b[3], b[5]))
        return out


def garble_program(ctx, key, steps, prim, rnd, window=64, intern=True, view=False, deferred=True):
    """returns (stream bytes as one array, per-step byte counts, seconds, stats, the Stream — still open); g.inputs0 =
    the zero labels of the primary inputs as they were BEFORE the program ran (a program may overwrite its inputs)"""
    L = engine.lib()
    g = engine.Stream(ctx, key, rnd, prim)
    g.inputs0 = [g.get(w)["l0"].copy() for w in prim]
    a = _Args(steps)
    n = len(steps)
    cap = sum(c.NumGates * 13 + c.slab_rows() * 16 for c, _, _ in steps) + 64
    out = np.empty(cap, np.uint8)
    base = out.ctypes.data
    sizes = np.zeros(n, np.int64)
    nb = C.c_size_t(0)
    pnb = C.byref(nb)
    # the bytes land in `out` — copied by the stream's copier threads while this thread queues on (gc_stream_garble_finish_async
    # + one gc_stream_garble_copies_wait at the end, inside the timed region), or by this thread (deferred=False)
    begin, finish, h = L.gc_stream_garble_begin, (L.gc_stream_garble_finish_async if deferred else L.gc_stream_garble_finish), g.h
    bargs = a.begin
    if intern:  # circuits named by handle (gc_stream_intern, once per circuit): no per-step content hash
        begin, bargs = L.gc_stream_garble_begin_h, a.interned(g)
    off = 0
    issued = 0
    if view:  # experiment: the bytes are handed out in place (gc_stream_garble_finish_view) and nobody reads them
        vptr = C.c_void_p(0)
        pv = C.byref(vptr)
        fview = L.gc_stream_garble_finish_view
    t0 = time.perf_counter()
    for k in range(n):
        lim = min(n, k + window)
        while issued < lim:
            rc = begin(h, *bargs[issued])
            if rc:
                raise engine.EngineError(rc, "gc_stream_garble_begin(step %d)" % issued)
            issued += 1
        rc = fview(h, pv, pnb) if view else finish(h, C.c_void_p(base + off), cap - off, pnb)
        if rc:
            raise engine.EngineError(rc, "gc_stream_garble_finish(step %d)" % k)
        sizes[k] = nb.value
        off += nb.value
    if deferred and not view:
        rc = L.gc_stream_garble_copies_wait(h)
        if rc:
            raise engine.EngineError(rc, "gc_stream_garble_copies_wait")
    dt = time.perf_counter() - t0
    return out[:off], sizes, dt, g.stats(), g


def eval_program(ctx, key, steps, prim, g, stream, sizes):
    """the evaluator alone over the produced bytes (all-zero inputs); returns seconds and checks the last outputs"""
    L = engine.lib()
    ev = engine.StreamEval(ctx, key)
    for w, l0 in zip(prim, g.inputs0):
        ev.set(w, (int(l0["d0"]), int(l0["d1"])))
    n = len(steps)
    call = L.gc_stream_eval_circuit
    used = C.c_size_t(0)
    pu = C.byref(used)
    base = stream.ctypes.data
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    args = [(c.NumGates, c.NumWires, max(max(i), max(o)) + 1) for c, i, o in steps]
    h = ev.h
    # blocks seen for the first time are parsed gate by gate and their circuits loaded (the peer's data: nothing can be
    # interned ahead, as the garbler does); "steady" is what follows the last such block
    pa, pm = C.c_uint64(0), C.c_uint64(0)
    ppa, ppm = C.byref(pa), C.byref(pm)
    parsed_before, steady_steps = 0, n
    t0 = t_known = time.perf_counter()
    for k in range(n):
        rc = call(h, args[k][0], args[k][1], args[k][2], C.c_void_p(base + int(offs[k])), int(sizes[k]), pu)
        if rc:
            raise engine.EngineError(rc, "gc_stream_eval_circuit(step %d)" % k)
        if used.value != sizes[k]:
            raise RuntimeError("evaluator consumed %d of %d bytes of step %d" % (used.value, sizes[k], k))
        L.gc_stream_eval_stats(h, ppa, ppm)
        if pa.value != parsed_before:
            parsed_before, steady_steps, t_known = pa.value, n - 1 - k, time.perf_counter()
    outs = steps[-1][2][:8]
    got = [ev.get(o) for o in outs]  # waits for everything
    dt = time.perf_counter() - t0
    eval_program.steady = (time.perf_counter() - t_known, steady_steps)
    for o, lab in zip(outs, got):  # valid labels of the garbler's wires
        wire = g.get(o)
        assert lab in ((int(wire["l0"]["d0"]), int(wire["l0"]["d1"])), (int(wire["l1"]["d0"]), int(wire["l1"]["d1"]))), o
    st = ev.stats()
    ev.close()
    return dt, st


def eval_program_blocks(ctx, key, steps, prim, g, stream, sizes, piece=32 << 20, pinned=True):
    """the evaluator over the FRAMED stream (20-byte OpCircuit headers + blocks, as the peer's p2p.Conn delivers it) through
    gc_stream_eval_blocks in pieces of `piece` bytes of a read buffer — pinned (gc_host_alloc: the DMA reads it in place) or
    pageable.  Whole read buffers go to the GPU, which recognises the blocks itself (mpc_amd/csrc/stream_eval_dev.cpp).
    Returns seconds, (parsed, matched); checks the last outputs against the garbler's wires."""
    import struct
    L = engine.lib()
    n = len(steps)
    total = int(sizes.sum()) + 20 * n + 4
    hold = engine.PinnedArray((total,), np.uint8) if pinned else None
    framed = hold.a if pinned else np.empty(total, np.uint8)
    pos, off = 0, 0
    for k, (c, i, o) in enumerate(steps):  # (outside the timed region: this is the peer's work)
        framed[pos:pos + 20] = np.frombuffer(struct.pack(">5I", 1, k & 0xffffffff, c.NumGates, c.NumWires, max(max(i), max(o)) + 1), np.uint8)
        framed[pos + 20:pos + 20 + sizes[k]] = stream[off:off + sizes[k]]
        pos += 20 + int(sizes[k])
        off += int(sizes[k])
    framed[pos:pos + 4] = np.frombuffer(struct.pack(">I", 2), np.uint8)
    ev = engine.StreamEval(ctx, key)
    for w, l0 in zip(prim, g.inputs0):
        ev.set(w, (int(l0["d0"]), int(l0["d1"])))
    call = L.gc_stream_eval_blocks
    used, nb, more = C.c_size_t(0), C.c_uint32(0), C.c_int(0)
    pu, pn, pm = C.byref(used), C.byref(nb), C.byref(more)
    base = framed.ctypes.data
    end = total - 4
    at, done, win = 0, 0, piece
    t0 = time.perf_counter()
    while done < n:
        rc = call(ev.h, C.c_void_p(base + at), min(win, total - at), pu, pn, pm)
        if rc:
            raise engine.EngineError(rc, "gc_stream_eval_blocks(at byte %d)" % at)
        if not used.value and not more.value:
            raise RuntimeError("gc_stream_eval_blocks stopped at byte %d of %d" % (at, total))
        win = piece if used.value else win * 2  # (a block longer than the piece: come back with more)
        at += used.value
        done += nb.value
    outs = steps[-1][2][:8]
    got = [ev.get(o) for o in outs]  # waits for everything
    dt = time.perf_counter() - t0
    assert at == end, (at, end)
    for o, lab in zip(outs, got):
        wire = g.get(o)
        assert lab in ((int(wire["l0"]["d0"]), int(wire["l0"]["d1"])), (int(wire["l1"]["d0"]), int(wire["l1"]["d1"]))), o
    st = ev.stats()
    fz = ev.fuse_stats() + ev.dev_stats() + (ev.wait_stats(),)
    ev.close()
    if hold is not None:
        hold.close()
    return dt, st, fz


def golden_sha(name, key):
    try:
        with open(GOLDEN) as f:
            return json.load(f).get("%s/key%d" % (name, len(key)))
    except (OSError, ValueError):
        return None


def run_program(name, key=bytes(range(32)), ctx=None, window=64, evaluate=True, repeats=2, intern=True, view=False):
    steps, prim = PROGRAMS[name]()
    rnd = stream_rnd(name, len(prim))
    own = ctx is None
    if own:
        ctx = engine.Context(0)
    gates = sum(c.NumGates for c, _, _ in steps)
    ands = sum(c.stats()["AND"] for c, _, _ in steps)
    best = None
    for rep in range(repeats):  # the first pass builds and caches the plans: report the second as the steady state
        stream, sizes, dt, stats, g = garble_program(ctx, key, steps, prim, rnd, window, intern, view)
        sha = hashlib.sha256(stream.tobytes() if len(stream) < (1 << 30) else memoryview(stream)).hexdigest()
        res = {"program": name, "steps": len(steps), "gates": gates, "and": ands, "window": window, "interned": intern,
               "garble_s": dt, "garble_gates_per_s": gates / dt, "garble_us_per_step": dt / len(steps) * 1e6,
               "stream_bytes": int(len(stream)), "launch_groups": stats[0], "grouped_steps": stats[1], "big_steps": stats[2],
               "deep_steps": g.deep_stats()[0], "lanes": g.deep_stats()[1], "sha256": sha}
        if rep == 0:
            res["first_pass_s"] = dt
            first = dt
        else:
            res["first_pass_s"] = first
        res["fuse"] = g.fuse_stats()
        res["waiting_units"] = g.wait_stats()
        want = golden_sha(name, key) if not view else None
        if want is not None and want != sha:
            raise AssertionError("%s: stream SHA-256 %s != oracle's %s" % (name, sha, want))
        if evaluate and rep == repeats - 1 and not view:
            # (twice, as the garbler: the merged plans of fused chains are cached per ctx — the second evaluator finds them)
            for _ in range(2):
                edt, est = eval_program(ctx, key, steps, prim, g, stream, sizes)
            res.update({"eval_s": edt, "eval_gates_per_s": gates / edt, "eval_us_per_step": edt / len(steps) * 1e6,
                        "eval_blocks_parsed": est[0], "eval_blocks_matched": est[1]})
            # ... and over the framed stream, whole read buffers at a time (the device recognises the blocks)
            for _ in range(2):
                bdt, bst, bfz = eval_program_blocks(ctx, key, steps, prim, g, stream, sizes)
            res.update({"eval_blocks_s": bdt, "eval_blocks_gates_per_s": gates / bdt, "eval_blocks_parsed": bst[0], "eval_blocks_matched": bst[1],
                        "eval_blocks_piece": 32 << 20, "eval_blocks_buffer": "pinned (gc_host_alloc)", "eval_fuse": list(bfz)})
            sdt, sn = eval_program.steady
            if sn:  # after the last block the evaluator had not seen before
                res.update({"eval_steady_us_per_step": sdt / sn * 1e6, "eval_steady_steps": sn,
                            "eval_steady_gates_per_s": gates / len(steps) * sn / sdt, "eval_first_blocks_s": edt - sdt})
        g.close()
        best = res
    want = golden_sha(name, key) if not view else None
    best["sha256_golden"] = want
    best["sha256_ok"] = None if want is None else bool(want == best["sha256"])
    if want is not None and want != best["sha256"]:
        raise AssertionError("%s: stream SHA-256 %s != oracle's %s" % (name, best["sha256"], want))
    if own:
        ctx.close()
    return best


NATIVE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stream_driver")


def write_program(path, key, rnd, prim, steps, window):
    """the program as tools/stream_driver.c reads it (see the format there)"""
    import struct
    # circuits by CONTENT: the big-step programs build an object per step of four distinct circuits (until round 6 the file
    # of big130 carried 991 gate lists: 2.6 GB through /tmp for 10 MB of circuits)
    circs, index, by_content = [], {}, {}
    for c, _, _ in steps:
        if id(c) not in index:
            g = np.ascontiguousarray(c.Gates, dtype=GATE)
            ck = (hashlib.sha256(g.tobytes()).digest(), c.NumWires, c.num_inputs, c.num_outputs)
            if ck not in by_content:
                by_content[ck] = len(circs)
                circs.append(c)
            index[id(c)] = by_content[ck]
    with open(path, "wb") as f:
        f.write(b"GCSP" + struct.pack("<I", 1))
        f.write(struct.pack("<I", len(key)) + bytes(key))
        f.write(struct.pack("<Q", len(rnd)) + bytes(rnd))
        f.write(struct.pack("<I", len(prim)) + np.asarray(prim, np.uint32).tobytes())
        f.write(struct.pack("<I", len(circs)))
        for c in circs:
            g = np.ascontiguousarray(c.Gates, dtype=GATE)
            assert g.dtype.itemsize == 20
            f.write(struct.pack("<IIII", len(g), c.NumWires, c.num_inputs, c.num_outputs) + g.tobytes())
        f.write(struct.pack("<I", len(steps)))
        for c, in_, out_ in steps:
            f.write(struct.pack("<I", index[id(c)]) + np.asarray(in_, np.uint32).tobytes() + np.asarray(out_, np.uint32).tobytes())
        f.write(struct.pack("<I", window))


def run_native_steps(steps, prim, rnd, key=bytes(range(32)), window=64, env=None, timeout=600):
    """tools/stream_driver over a program given as steps: the driver's own JSON line (it has checked its six passes against one
    another).  A non-zero exit raises with what the child wrote to stderr."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "program.bin")
        write_program(path, key, rnd, prim, steps, window)
        full = dict(os.environ)
        full.update(env or {})
        run = subprocess.run([NATIVE, path], capture_output=True, text=True, timeout=timeout, env=full)
        if run.returncode != 0:  # keep what the child said: the exit status alone explains nothing (BENCH_r05 native_host.big130)
            raise RuntimeError("tools/stream_driver exit %d: %s" % (run.returncode, (run.stderr or run.stdout).strip()[-600:]))
        if os.environ.get("GC_TRACE"):  # the engine's wall-clock laps (developer aid): pass them on
            sys.stderr.write(run.stderr)
    r = json.loads(run.stdout.strip().splitlines()[-1])
    r["stderr"] = run.stderr[-2000:]
    return r


def run_native(name, key=bytes(range(32)), window=64, env=None):
    """the same program driven by tools/stream_driver (plain C over the C ABI: what a cgo host sees, no interpreter
    between the calls); the SHA-256 of its byte stream is checked against the oracle's like the Python-driven run's.
    None if the driver has not been built."""
    if not os.path.exists(NATIVE):
        return None
    steps, prim = PROGRAMS[name]()
    rnd = stream_rnd(name, len(prim))
    gates = sum(c.NumGates for c, _, _ in steps)
    r = run_native_steps(steps, prim, rnd, key, window, env=env)
    want = golden_sha(name, key)
    if want is not None and want != r["sha256"]:
        raise AssertionError("%s (native driver): stream SHA-256 %s != oracle's %s" % (name, r["sha256"], want))
    return {"program": name, "host": "tools/stream_driver (C)", "steps": r["steps"], "window": window, "gates": gates,
            "garble_gates_per_s": gates / r["garble_s"], "garble_us_per_step": r["garble_s"] / r["steps"] * 1e6,
            "eval_gates_per_s": gates / r["eval_s"], "eval_us_per_step": r["eval_s"] / r["steps"] * 1e6,
            # the framed stream through gc_stream_eval_blocks in pieces of a p2p.Conn read buffer (a helper thread compares ahead)
            "eval_blocks_gates_per_s": gates / r["eval_blocks_s"] if r.get("eval_blocks_s") else None,
            # ... the bytes handed out in place (no copy into a second buffer) / read buffers of 32 MiB in pinned memory
            "garble_view_gates_per_s": gates / r["garble_view_s"] if r.get("garble_view_s") else None,
            "garble_async_gates_per_s": gates / r["garble_async_s"] if r.get("garble_async_s") else None,
            "eval_blocks_pinned_gates_per_s": gates / r["eval_blocks_pinned_s"] if r.get("eval_blocks_pinned_s") else None,
            "eval_blocks_chunk": r.get("eval_blocks_chunk"),
            "eval_blocks_matched": r["eval_blocks_matched"], "sha256": r["sha256"],
            "eval_steady_us_per_step": r["eval_steady_s"] / max(r["eval_steady_steps"], 1) * 1e6,
            "eval_steady_gates_per_s": gates / r["steps"] * r["eval_steady_steps"] / max(r["eval_steady_s"], 1e-9),
            "eval_first_blocks_s": r["eval_s"] - r["eval_steady_s"], "eval_steady_steps": r["eval_steady_steps"],
            "sha256_ok": None if want is None else True}


def run_for_line(key=bytes(range(32)), ctx=None):
    """the `stream` object of bench.py's line: the big-step program at config 5's size (991 steps of 131 072 gates =
    1.3e8 gates, 2.4 GB of stream: about 2 s for both sides and both hosts), the two uniform small-step programs and the
    mixed program and the 23-circuit instruction mix (ssa23); every SHA-256 checked against the oracle's"""
    out = {}
    wall, t_last = {}, [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        wall[name] = round(now - t_last[0], 2)
        t_last[0] = now
    b = run_program("big130", key, ctx, window=4)  # (an interpreter's jitter between the calls: 2 in flight are enough for a C host)
    out.update({"program": "big130", "steps": b["steps"], "gates": b["gates"], "window": b["window"], "steady_ms_per_step": b["garble_us_per_step"] / 1e3,
                "steady_gates_per_s": b["garble_gates_per_s"],
                # the evaluator cannot intern: the first block of each circuit is parsed and loaded inside the timed run
                "eval_ms_per_step_all": b["eval_us_per_step"] / 1e3, "eval_gates_per_s_all": b["eval_gates_per_s"],
                "eval_steady_ms_per_step": b.get("eval_steady_us_per_step", b["eval_us_per_step"]) / 1e3,
                "eval_steady_gates_per_s": b.get("eval_steady_gates_per_s", b["eval_gates_per_s"]),
                "eval_first_blocks_s": b.get("eval_first_blocks_s"), "eval_blocks_parsed": b["eval_blocks_parsed"],
                "first_pass_s": b["first_pass_s"], "sha256": b["sha256"], "sha256_ok": b["sha256_ok"]})
    out["shape"] = SHAPES["big130"]
    lap("big130")
    for name in ("ed25519like", "ssa23", "mixed", "uniform512", "uniform4096"):
        r = run_program(name, key, ctx, window=WINDOWS.get(name, 64))
        out[name] = {k: r[k] for k in ("steps", "gates", "and", "window", "garble_gates_per_s", "garble_us_per_step", "eval_gates_per_s",
                                       "eval_us_per_step", "eval_steady_gates_per_s", "eval_blocks_gates_per_s", "eval_blocks_buffer",
                                       "eval_blocks_piece", "eval_fuse", "launch_groups", "grouped_steps", "big_steps",
                                       "deep_steps", "lanes", "sha256", "sha256_ok") if k in r}
        out[name]["shape"] = SHAPES[name]
        lap(name)
    # The UNCHANGED caller: compiler/ssa/streamer.go:694 calls Streaming.Garble one instruction at a time —
    # begin + finish per step, nothing queued ahead (window 1).  Every row above queues 64 - 1 024 instructions ahead, which
    # needs the three-edit patch of the streamer (go/ssa/stream_window_hip.go).
    r = run_program("ed25519like", key, ctx, window=1, evaluate=False)
    out["ed25519like_window1"] = {k: r[k] for k in ("steps", "gates", "window", "garble_gates_per_s", "garble_us_per_step", "launch_groups",
                                                    "sha256", "sha256_ok") if k in r}
    out["ed25519like_window1"]["caller"] = "unchanged: Streaming.Garble per instruction (gc_stream_garble_begin + _finish, nothing in flight)"
    lap("ed25519like_window1")
    # ... and the engine's rate when the bytes are consumed IN PLACE (gc_stream_garble_finish_view, as go/circuit/stream_hip.go
    # does: a pointer into the engine's pinned staging, no copy into a second buffer)
    r = run_program("ed25519like", key, ctx, window=WINDOWS["ed25519like"], view=True)
    out["ed25519like"]["garble_view_gates_per_s"] = r["garble_gates_per_s"]
    out["ed25519like"]["fuse"] = dict(zip(("fused_units", "fused_steps", "plans_asked", "unfit"), r.get("fuse", ())))
    out["ed25519like"]["fuse"]["waiting_units"] = r.get("waiting_units")
    # the same programs with a C host in place of this interpreter (what a cgo caller gets)
    native = {}
    for name, win in (("big130", 2), ("ed25519like", WINDOWS["ed25519like"]), ("uniform512", 64), ("uniform4096", 64), ("mixed", 64),
                      ("ssa23", 64)):
        try:
            r = run_native(name, key, win)
        except Exception as e:  # a side measurement: reported, never fatal for the bench line — with what the child said, and once more
            first = str(e)[:800]
            try:
                r = run_native(name, key, win)
                r["first_attempt_error"] = first
            except Exception as e2:
                r = {"error": str(e2)[:800], "first_attempt_error": first}
        if r is not None:
            native[name] = {k: r[k] for k in r if k in ("garble_gates_per_s", "garble_view_gates_per_s", "garble_async_gates_per_s", "garble_us_per_step", "eval_gates_per_s", "eval_blocks_gates_per_s", "eval_blocks_pinned_gates_per_s",
                                                      "eval_us_per_step", "eval_steady_gates_per_s", "eval_steady_us_per_step",
                                                      "window", "sha256_ok", "error", "first_attempt_error")}
    lap("ed25519like view + native_host")
    # The round-6 EXPERIMENT (off in every row above): launch units ordered by per-wire versions on the device and run by persistent
    # workgroups that claim published units in program order (GC_STREAM_DATAFLOW=4; needs 16 hardware queues) — C host, its own process
    exp = {}
    for label, name, win, og in (("ssa23_window256", "ssa23", 256, None), ("ssa23_window64_eager", "ssa23", 64, "1")):
        env = {"GC_STREAM_DATAFLOW": "4", "GPU_MAX_HW_QUEUES": "16"}
        if og:
            env["GC_STREAM_OPEN_GROUPS"] = og
        try:
            r = run_native(name, key, win, env=env)
            exp[label] = {k: r[k] for k in ("window", "garble_gates_per_s", "garble_view_gates_per_s", "sha256_ok")}
        except Exception as e:  # an experiment's side row: reported, never fatal
            exp[label] = {"error": str(e)[:400]}
    exp["note"] = ("EXPERIMENT, off by default (DESIGN.md §5, EXPERIMENTS.md round 6): ssa23 garbling under GC_STREAM_DATAFLOW=4; the rows "
                   "above it are the product's default; it costs ed25519like 17 %")
    out["dataflow_experiment"] = exp
    lap("dataflow_experiment")
    out["wall_s"] = wall
    if native:
        out["native_host"] = native
    out["published_reference"] = "1.4e7 gates/s, Go, i5-8257U (benchmarks.md:677-704: Ed25519 sign.mpcl streamed)"
    return out


def run(total_gates=10_000_000, key=bytes(range(32)), ctx=None, evaluate=True):
    """compatibility with round 2's callers: the big-step program of `total_gates` gates"""
    PROGRAMS["_big"] = lambda: program_big(total_gates)
    r = run_program("_big", key, ctx, window=2, evaluate=evaluate)
    return r


if __name__ == "__main__":
    names = sys.argv[1:] or ["big", "uniform512", "uniform4096", "mixed"]
    for nm in names:  # name[:window[:noh]]   noh: per-step content look-up (gc_stream_garble_begin) instead of handles
        parts = nm.split(":")
        nm = parts[0]
        win = int(parts[1]) if len(parts) > 1 else 64
        if len(parts) > 2 and parts[2] == "native":
            print(json.dumps(run_native(nm, window=2 if nm.startswith("big") and win == 64 else win)), flush=True)
            continue
        print(json.dumps(run_program(nm, window=2 if nm.startswith("big") and win == 64 else win,
                                     intern=not (len(parts) > 2 and parts[2] == "noh"), view=len(parts) > 2 and parts[2] == "view")), flush=True)
