#!/usr/bin/env python3
"""Hostile bytes for the streaming evaluator: gc_stream_eval_circuit parses the PEER's data, and its fast
path decides from untrusted bytes which bytes to skip (the byte-skeleton matcher + helper threads of stream_engine.cpp).
Mutational fuzz over recorded valid OpCircuit blocks of scripts-style programs:

  bit flips anywhere · truncation · the id-width flag 0x10 flipped · tmp flags 0x80 / 0x40 / 0x20 flipped · ids swapped inside a
  gate · an id replaced by another id of the block (breaks the repeat pattern of the global ids) or by one out of range ·
  table-row bytes changed only · the op nibble changed (valid and invalid) · the header's gate count off by a few

Every mutant goes to an evaluator that has SEEN the valid block (so the skeleton path is tried first, the gate-by-gate parser
when it does not match), with GC_STREAM_THREADS 0 and 3 for the big blocks, and to the oracle's restated StreamEval
(oracle/stream_oracle.c, circuit/stream_evaluator.go:271-432) started from the same wire store.  Required:

  * no crash, no hang;
  * ACCEPTED by the engine  =>  the oracle accepts it too, consumes the same number of bytes and ends with the same labels on
    every wire the block names (and on a sample of the others);
  * REJECTED by the engine  =>  the oracle rejects it (same class: truncated / unknown operation), or the block is one of the
    cases where the engine is stricter than the reference's loop by design — a global wire id beyond the block's numWires
    (the reference indexes its slice with it: a panic there), a tmp id beyond numTmpWires, a tmp wire read before the block
    wrote it (the reference would hand out a label of an EARLIER circuit), more gates announced than the bytes can hold —
    and the engine's wire store is exactly what it was before the block; the stream stays usable.

usage: tests/hostile_fuzz.py [mutants [seed]]   (one summary line per 500 mutants; profiles/r04_hostile_fuzz.log)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np

if os.environ.get("GC_FUZZ_DEFAULT_PLANNER"):  # the product's default instead: plans in the background, unplanned chains step by step
    os.environ.pop("GC_STREAM_FUSE_EAGER", None)
else:
    os.environ.setdefault("GC_STREAM_FUSE_EAGER", "1")  # every chain on its merged plan, at first sight (as tests/conftest.py)

import oracle
from mpc_amd import engine
from tests.util import drbg

ROWS = {0: 0, 1: 0, 2: 2, 3: 3, 4: 1}  # XOR XNOR AND OR INV


def parse(block, ngates):
    """the reference's reading of a block: [(pos of the op byte, op, flags, short, [ids], [pos of ids], pos of rows, nrows)],
    or the error class ("rows" / "gate") where it stops"""
    gates, pos = [], 0
    for _ in range(ngates):
        if pos + 1 > len(block):
            return gates, "rows"
        gop = block[pos]
        flags, short, op = gop & 0xe0, bool(gop & 0x10), gop & 0x0f
        p0 = pos
        pos += 1
        if op > 4:
            return gates, "gate"
        nw = 2 if op == 4 else 3
        sz = 2 if short else 4
        if pos + nw * sz + 16 * ROWS[op] > len(block):
            return gates, "rows"
        ids, idpos = [], []
        for i in range(nw):
            idpos.append(pos)
            ids.append(int.from_bytes(block[pos:pos + sz], "big"))
            pos += sz
        gates.append((p0, op, flags, short, ids, idpos, pos, ROWS[op]))
        pos += 16 * ROWS[op]
    return gates, None


def stricter(block, ngates, ntmp, nwires):
    """is the block one the engine refuses by design although the reference's loop would walk it?"""
    if ngates > len(block) // 5:
        return "more gates than bytes"
    gates, err = parse(block, ngates)
    written = set()
    for p0, op, flags, short, ids, idpos, rpos, nrows in gates:
        ins = [(ids[0], flags & 0x80)] + ([(ids[1], flags & 0x40)] if op != 4 else [])
        for v, t in ins:
            if t:
                if v >= ntmp:
                    return "tmp id out of range"
                if v not in written:
                    return "tmp read before written"
            elif v >= nwires:
                return "global id out of range"
        c = ids[-1]
        if flags & 0x20:
            if c >= ntmp:
                return "tmp id out of range"
            written.add(c)
        elif c >= nwires:
            return "global id out of range"
    return None


def mutate(rng, block, ngates):
    """one mutant: (bytes, gate count for the header, what was done)"""
    b = bytearray(block)
    gates, _ = parse(block, ngates)
    kind = int(rng.integers(0, 11))
    g = gates[int(rng.integers(0, len(gates)))]
    p0, op, flags, short, ids, idpos, rpos, nrows = g
    sz = 2 if short else 4
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            i = int(rng.integers(0, len(b)))
            b[i] ^= 1 << int(rng.integers(0, 8))
        return bytes(b), ngates, "bit flips"
    if kind == 1:
        return bytes(b[:int(rng.integers(0, len(b)))]), ngates, "truncated"
    if kind == 2:
        b[p0] ^= 0x10
        return bytes(b), ngates, "id width flipped"
    if kind == 3:
        b[p0] ^= int(rng.choice([0x80, 0x40, 0x20]))
        return bytes(b), ngates, "tmp flag flipped"
    if kind == 4 and len(ids) == 3:
        i, j = rng.choice(3, 2, replace=False)
        vi, vj = b[idpos[i]:idpos[i] + sz], b[idpos[j]:idpos[j] + sz]
        b[idpos[i]:idpos[i] + sz], b[idpos[j]:idpos[j] + sz] = vj, vi
        return bytes(b), ngates, "ids swapped"
    if kind == 5:
        other = gates[int(rng.integers(0, len(gates)))]
        v = other[4][int(rng.integers(0, len(other[4])))]
        i = int(rng.integers(0, len(ids)))
        b[idpos[i]:idpos[i] + sz] = (v & ((1 << (8 * sz)) - 1)).to_bytes(sz, "big")
        return bytes(b), ngates, "id replaced by another of the block"
    if kind == 6:
        i = int(rng.integers(0, len(ids)))
        v = int(rng.choice([0xffff, 0xfffe, 0x7fff])) if short else int(rng.choice([0xffffffff, 0x80000000, 0x00ffffff, 0x10000]))
        b[idpos[i]:idpos[i] + sz] = v.to_bytes(sz, "big")
        return bytes(b), ngates, "id out of range"
    if kind == 7:
        rowed = [q for q in gates if q[7]]
        if rowed:
            q = rowed[int(rng.integers(0, len(rowed)))]
            for _ in range(int(rng.integers(1, 5))):
                b[q[6] + int(rng.integers(0, 16 * q[7]))] ^= int(rng.integers(1, 256))
            return bytes(b), ngates, "row bytes only"
    if kind == 8:
        b[p0] = (b[p0] & 0xf0) | int(rng.integers(0, 16))
        return bytes(b), ngates, "op nibble changed"
    if kind == 9:
        return bytes(b), max(1, ngates + int(rng.integers(-3, 4))), "gate count off"
    i = int(rng.integers(0, len(b)))
    b[i] = int(rng.integers(0, 256))
    return bytes(b), ngates, "byte replaced"


def programs(big=False, deep=False):
    """valid blocks to mutate: small SSA-step circuits of every gate type (tmp wires, 16- and 32-bit ids, repeated ids); with
    big=True two blocks of ~49 000 gates whose skeletons are matched in segments by the helper threads; with deep=True blocks
    that run on the evaluator's deep lanes (GC_STREAM_DEEP_STEPS=100: a 128-bit adder and subtractor — rows through the group's
    upload region — and a 128-bit multiplier of 56 140 gates, whose rows go up from the pinned ring into the slot's arena)"""
    from mpc_amd.circuit import adder, comparator64, multiplier, subtractor, synthetic_levelised
    shapes = [synthetic_levelised(6, 40, 0.3, seed=51, ninputs=24, inv_frac=0.1, xnor_frac=0.1),
              synthetic_levelised(3, 60, 0.4, seed=52, ninputs=24, or_frac=0.15, inv_frac=0.05),
              adder(16), comparator64(), synthetic_levelised(10, 64, 0.25, seed=57, ninputs=64)]
    if big:
        shapes = [synthetic_levelised(24, 2048, 0.3, seed=58, ninputs=256, inv_frac=0.05)]
    if deep:
        shapes = [adder(128), subtractor(128), multiplier(128)]
    out = []
    for k, c in enumerate(shapes):
        for base in (0, 0x11000):  # short and long id forms
            n = c.num_inputs
            ins = [base + (7 * k + i) % 300 for i in range(n)]
            if k % 2:
                ins[1] = ins[0]  # a repeated global id
            outs = [base + 400 + i for i in range(c.num_outputs)]
            out.append((c, ins, outs))
    return out


def run(mutants=200, seed=1, big=False, log=None, deep=False, framed=False):
    """returns (mutants run, accepted, rejected, by kind) — raises AssertionError on the first violation.
    framed: every mutant goes through gc_stream_eval_blocks instead — the 20-byte header (OpCircuit, step, numGates,
    numTmpWires, numWires) in front, an OpReturn word behind — and one mutant in six also has a header size changed (0, off by
    a few, 2^31, 2^32 - 1: numTmpWires and numWires size arrays)"""
    import struct
    rng = np.random.default_rng(seed)
    key = drbg("hostile-key", 32)
    ctx = engine.Context(0)
    if deep:
        os.environ["GC_STREAM_DEEP_STEPS"] = "100"
    steps = programs(big, deep)
    prim = sorted({w for _, i, _ in steps for w in i})
    rnd = drbg("hostile-rnd", 16 * (len(prim) + 1))
    og = oracle.Stream(key, rnd, prim)
    blocks = []
    for c, in_, out_ in steps:
        data = og.garble(c.Gates, c.NumWires, in_, out_)
        ntmp = c.NumWires
        nw = max(max(in_), max(out_)) + 1
        blocks.append((data, c.NumGates, ntmp, nw, sorted(set(in_) | set(out_))))
    ge = engine.StreamEval(ctx, key)
    model = {}  # wire -> label: what both stores hold
    for w in prim:
        model[w] = (int(og.get(w)["l0"]["d0"]), int(og.get(w)["l0"]["d1"]))
        ge.set(w, model[w])

    def oracle_on(block, ngates, ntmp, nw, wires):
        oe = oracle.StreamEval(key)
        for w, l in model.items():
            oe.set(w, l)
        try:
            used = oe.circuit(ngates, ntmp, nw, block)
        except oracle.OracleError as e:
            return e.code, None, None
        after = {}
        for w in wires:
            try:
                after[w] = oe.get(w)
            except oracle.OracleError:
                after[w] = (0, 0)
        return 0, used, after

    # the evaluator sees every valid block first: its skeleton is recorded, the mutants meet the fast path
    for data, ng, ntmp, nw, wires in blocks:
        rc, used, after = oracle_on(data, ng, ntmp, nw, wires)
        assert rc == 0 and ge.circuit(ng, ntmp, nw, data) == used == len(data)
        model.update(after)
    stats = {"accepted": 0, "rejected": 0, "rejected_stricter": 0}
    kinds = {}
    for m in range(mutants):
        data, ng, ntmp, nw, wires = blocks[int(rng.integers(0, len(blocks)))]
        mut, mng, what = mutate(rng, data, ng)
        named = set(wires)
        for q in parse(mut, mng)[0]:
            named |= {v for v in q[4] if v < nw + 64}
        named = sorted(w for w in named if w < (1 << 22))
        tail = b""
        if framed:
            tail = struct.pack(">I", 2) + bytes(16)  # OpReturn: the call stops in front of it
            if rng.integers(0, 6) == 0:
                if rng.integers(0, 2):
                    ntmp = int(rng.choice([0, 1, max(ntmp - 1, 0), ntmp + 7, 64 * mng + (1 << 20) + 1, 1 << 31, 0xffffffff]))
                    what += " + numTmpWires"
                else:
                    nw = int(rng.choice([0, max(nw - 1, 0), nw + 1000, (1 << 28) + 1, 0xffffffff]))
                    what += " + numWires"
        sized = ntmp > 64 * mng + (1 << 20) or nw > (1 << 28)  # refused by the engine before anything is sized (GC_E_ARG)
        named = [w for w in named if w < nw]
        # (the reference reads on from the connection: what follows the mutant belongs to the bytes both sides see)
        orc, oused, oafter = (-5, None, None) if sized else oracle_on(mut + tail, mng, ntmp, nw, named)
        try:
            if framed:
                used, nb, more = ge.blocks(struct.pack(">5I", 1, m, mng, ntmp, nw) + mut + tail)
                if more and nb == 0:
                    erc, used = engine.GC_E_ROWS, None  # (a block that ends beyond the buffer: the per-block call's truncation)
                elif nb != 1 or more or used - 20 != oused:
                    # the mutant ends elsewhere than its bytes and what follows reads as another operation word: no reference
                    # behaviour to compare with beyond "the first block was taken as the oracle takes it"
                    assert nb >= 1 and orc == 0, "mutant %d (%s): blocks() = %s, oracle %s" % (m, what, (used, nb, more), (orc, oused))
                    kinds[what + " (ran on)"] = kinds.get(what + " (ran on)", 0) + 1
                    for w in named:
                        if w in oafter:
                            model[w] = ge.get(w)  # (a second pseudo-block may have written it)
                    continue
                else:
                    used, erc = used - 20, 0
            else:
                used = ge.circuit(mng, ntmp, nw, mut)
                erc = 0
        except engine.EngineError as e:
            erc, used = e.code, None
            if framed and ge.last_blocks[1] > 0:  # the mutant itself was taken; the error is of the bytes behind it
                erc, used = 0, oused
        if sized:
            assert erc == engine.GC_E_ARG, "mutant %d (%s): header sizes beyond the bounds were not refused (%d)" % (m, what, erc)
        kinds[what] = kinds.get(what, 0) + 1
        if erc == 0:
            stats["accepted"] += 1
            assert orc == 0, "mutant %d (%s): the engine accepted what the oracle rejects (%d)" % (m, what, orc)
            assert used == oused, "mutant %d (%s): consumed %d, oracle %d" % (m, what, used, oused)
            for w in named:
                if w in oafter and w < nw:
                    got = ge.get(w)
                    assert got == oafter[w], "mutant %d (%s): wire %d differs from the oracle's" % (m, what, w)
                    model[w] = oafter[w]
        else:
            why = "header sizes" if sized else stricter(mut + tail, mng, ntmp, nw)
            if orc != 0:
                stats["rejected"] += 1
            else:
                assert why, "mutant %d (%s): the engine rejects (%d) what the reference's loop walks" % (m, what, erc)
                stats["rejected_stricter"] += 1
            # nothing of a rejected block may have reached the store
            for w in list(rng.choice(named, min(len(named), 6), replace=False)) if named else []:
                w = int(w)
                if w in model and w < nw:
                    assert ge.get(w) == model[w], "mutant %d (%s): a rejected block changed wire %d" % (m, what, w)
        if m % 50 == 49:  # the stream is still usable: a valid block evaluates to the oracle's labels
            data, ng, ntmp, nw, wires = blocks[int(rng.integers(0, len(blocks)))]
            rc, oused, oafter = oracle_on(data, ng, ntmp, nw, wires)
            assert rc == 0 and ge.circuit(ng, ntmp, nw, data) == oused
            for w in wires[::3]:
                assert ge.get(w) == oafter[w], "after mutant %d: a valid block no longer evaluates to the oracle's labels" % m
            model.update(oafter)
        if log and m % 500 == 499:
            log("%6d mutants: %s  parsed / matched blocks %s" % (m + 1, stats, ge.stats()))
    parsed, matched = ge.stats()
    if deep:
        assert ge.deep_stats()[0] > 0 or os.environ.get("GC_STREAM_DEEP_LANES") == "0", "no block ran on a deep lane"
        os.environ.pop("GC_STREAM_DEEP_STEPS", None)
    ge.close()
    ctx.close()
    return mutants, stats, kinds, (parsed, matched)


def run_device(mutants=100, seed=1, log=None, pinned=False):
    """The same mutants INSIDE read buffers that the device recognises (mpc_amd/csrc/stream_eval_dev.cpp: buffers of 64 KiB and
    more through gc_stream_eval_blocks): a few valid blocks the evaluator knows, the mutant, two more valid blocks, an OpReturn
    word.  The host only guesses where blocks are, the GPU says whether a guess holds and which repeat pattern the ids have:
    a mutant must be caught there or fall through to the host's parser.  Required: the valid blocks in front of the mutant are
    taken; the mutant is taken iff the oracle takes it (or refused by design: `stricter`), with the oracle's byte count; the
    blocks behind an accepted mutant are taken too; every wire the buffer names ends on the oracle's label."""
    import struct
    from mpc_amd.circuit import multiplier
    rng = np.random.default_rng(seed)
    key = drbg("hostile-key", 32)
    ctx = engine.Context(0)
    steps = programs()
    mul = multiplier(32)  # ~70 KB of block: a buffer with one of these is one the device gets
    for base in (0, 0x11000):
        steps.append((mul, [base + 500 + i for i in range(64)], [base + 600 + i for i in range(32)]))
    prim = sorted({w for _, i, _ in steps for w in i})
    rnd = drbg("hostile-rnd", 16 * (len(prim) + 1))
    og = oracle.Stream(key, rnd, prim)
    blocks = []
    for c, in_, out_ in steps:
        data = og.garble(c.Gates, c.NumWires, in_, out_)
        blocks.append((data, c.NumGates, c.NumWires, max(max(in_), max(out_)) + 1, sorted(set(in_) | set(out_))))
    big_ones = [k for k, b in enumerate(blocks) if len(b[0]) > 40000]
    every = sorted({w for b in blocks for w in b[4]})
    ge = engine.StreamEval(ctx, key)
    model = {}
    for w in prim:
        model[w] = (int(og.get(w)["l0"]["d0"]), int(og.get(w)["l0"]["d1"]))
        ge.set(w, model[w])

    def fresh_oracle():
        oe = oracle.StreamEval(key)
        for w, l in model.items():
            oe.set(w, l)
        return oe

    oe = fresh_oracle()
    for data, ng, ntmp, nw, wires in blocks:  # the evaluator sees every valid block first
        assert ge.circuit(ng, ntmp, nw, data) == oe.circuit(ng, ntmp, nw, data) == len(data)
    for w in every:
        model[w] = oe.get(w)
    hold = engine.PinnedArray((1 << 22,), np.uint8) if pinned else None
    stats = {"accepted": 0, "rejected": 0, "rejected_stricter": 0, "ran_on": 0}
    kinds = {}
    for m in range(mutants):
        pre = [int(rng.integers(0, len(blocks))) for _ in range(int(rng.integers(3, 7)))]
        pre.insert(int(rng.integers(0, len(pre) + 1)), big_ones[int(rng.integers(0, len(big_ones)))])
        if rng.integers(0, 3) == 0:
            pre.append(big_ones[int(rng.integers(0, len(big_ones)))])
        post = [int(rng.integers(0, len(blocks))) for _ in range(2)]
        mi = int(rng.integers(0, len(blocks)))
        data, ng, ntmp, nw, wires = blocks[mi]
        mut, mng, what = mutate(rng, data, ng)
        mnt, mnw = ntmp, nw
        if rng.integers(0, 8) == 0:
            if rng.integers(0, 2):
                mnt = int(rng.choice([0, 1, max(ntmp - 1, 0), ntmp + 7, 64 * mng + (1 << 20) + 1, 1 << 31, 0xffffffff]))
                what += " + numTmpWires"
            else:
                mnw = int(rng.choice([0, max(nw - 1, 0), nw + 1000, (1 << 28) + 1, 0xffffffff]))
                what += " + numWires"
        sized = mnt > 64 * mng + (1 << 20) or mnw > (1 << 28)
        frame = lambda k, b: struct.pack(">5I", 1, k, b[1], b[2], b[3]) + bytes(b[0])
        buf = b"".join(frame(k, blocks[k]) for k in pre)
        j = len(pre)
        buf += struct.pack(">5I", 1, m, mng, mnt, mnw) + mut
        buf += b"".join(frame(k, blocks[k]) for k in post) + struct.pack(">I", 2) + bytes(16)
        # the oracle: block after block; where does it stop?
        oe = fresh_oracle()
        for k in pre:
            b = blocks[k]
            assert oe.circuit(b[1], b[2], b[3], b[0]) == len(b[0])
        orc, oused = 0, None
        # (the reference reads on from the connection: what follows the mutant belongs to the bytes both sides see)
        following = buf[len(buf) - 20 - sum(len(blocks[k][0]) + 20 for k in post):]
        if sized:
            orc = -5
        else:
            try:
                oused = oe.circuit(mng, mnt, mnw, mut + following)
            except oracle.OracleError as e:
                orc = e.code
        try:
            if pinned:
                hold.a[:len(buf)] = np.frombuffer(buf, np.uint8)
                used, nb, more = ge.blocks_at(hold.a.ctypes.data, len(buf))
            else:
                used, nb, more = ge.blocks(buf)
            erc = 0
        except engine.EngineError as e:
            erc = e.code
            used, nb, more = ge.last_blocks
        kinds[what] = kinds.get(what, 0) + 1
        assert nb >= j, "mutant %d (%s): only %d of the %d valid blocks in front of it were taken (%d)" % (m, what, nb, j, erc)
        took = nb > j
        if sized:
            assert not took and erc == engine.GC_E_ARG, "mutant %d (%s): header sizes beyond the bounds were not refused" % (m, what)
        ran_on = took and (orc != 0 or oused != len(mut))
        if took and not ran_on:
            stats["accepted"] += 1
            for k in post:
                b = blocks[k]
                assert oe.circuit(b[1], b[2], b[3], b[0]) == len(b[0])
            assert nb == j + 3 and erc == 0, "mutant %d (%s): %d blocks taken of %d, status %d" % (m, what, nb, j + 3, erc)
        elif ran_on:
            # the mutant ends elsewhere than its bytes and what follows read as further operations: no reference behaviour to
            # compare with beyond "it was taken as the oracle takes it" — both stores start over from the engine's
            assert orc == 0, "mutant %d (%s): the engine took what the oracle rejects (%d)" % (m, what, orc)
            stats["ran_on"] += 1
            for w in every:
                model[w] = ge.get(w)
            continue
        else:
            if orc != 0:
                stats["rejected"] += 1
            else:
                why = "header sizes" if sized else stricter(mut + following, mng, mnt, mnw)
                cut = erc == 0 and more  # (the block "ends beyond the buffer": only if it really is longer than its bytes)
                assert why or cut, "mutant %d (%s): the engine rejects (%d) what the reference's loop walks" % (m, what, erc)
                stats["rejected_stricter"] += 1
            # nothing of a refused block may have reached the store (the reference's loop stops half-way through such a block:
            # the comparison is with the blocks in front of it)
            oe = fresh_oracle()
            for k in pre:
                b = blocks[k]
                oe.circuit(b[1], b[2], b[3], b[0])
        named = [w for w in every if w < (1 << 22)]
        for w in named[:: max(1, len(named) // 60)] + [w for w in wires if w < mnw]:
            assert ge.get(w) == oe.get(w), "mutant %d (%s): wire %d differs from the oracle's (taken %s, status %d, oracle %d, blocks %d of %d + 3, numWires %d of %d, block %d pre %s post %s)" % (
                m, what, w, took, erc, orc, nb, j, mnw, nw, mi, pre, post)
        for w in every:
            model[w] = oe.get(w)
        if log and m % 100 == 99:
            log("%6d mutants in device-matched buffers: %s  device blocks / fallbacks %s" % (m + 1, stats, ge.dev_stats()))
    dev = ge.dev_stats()
    ge.close()
    if hold is not None:
        hold.close()
    ctx.close()
    return mutants, stats, kinds, dev


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for threads in ("0", "3"):
        os.environ["GC_STREAM_THREADS"] = threads
        say = lambda s: print("threads=%s %s" % (threads, s), flush=True)
        r = run(n // 2, seed, big=False, log=say)
        print("threads=%s small blocks: %d mutants, %s, kinds %s, parsed/matched %s — no violation" % ((threads,) + r), flush=True)
        r = run(max(50, n // 100), seed + 1, big=True, log=say)
        print("threads=%s big blocks (helper-thread skeleton match): %d mutants, %s, kinds %s, parsed/matched %s — no violation" % ((threads,) + r), flush=True)
        r = run(max(50, n // 50), seed + 2, deep=True, log=say)
        print("threads=%s deep blocks (lanes; rows through the upload region / from the pinned ring): %d mutants, %s, kinds %s, parsed/matched %s — no violation" % ((threads,) + r), flush=True)
        r = run(n // 4, seed + 3, framed=True, log=say)
        print("threads=%s framed blocks through gc_stream_eval_blocks (header sizes mutated too): %d mutants, %s, kinds %s, parsed/matched %s — no violation" % ((threads,) + r), flush=True)
    for pinned in (False, True):
        r = run_device(n // 4, seed + 4, log=print, pinned=pinned)
        print("read buffers matched on the device (%s): %d mutants, %s, kinds %s, device blocks / fallbacks %s — no violation" % (
            ("pinned" if pinned else "pageable",) + r), flush=True)
