"""The C host (tools/stream_driver.c: plain C over include/gcengine.h, the calls a cgo shim makes) and the Python binding on
BIG streamed steps — the size class of BASELINE config 5 that BENCH_r05 saw fail from the C host (`native_host.big130`: exit 1,
stderr lost) while no `-m gpu` test drove that path.  Every byte against the oracle.

Reference: circuit/stream_garble.go:161-192,391-446 (Streaming.Garble + wire format), circuit/stream_evaluator.go:226-432."""
import hashlib
import os
import struct

import numpy as np
import pytest

import oracle
from mpc_amd import engine
from tests.util import drbg

pytestmark = pytest.mark.gpu


def _big_program(nsteps, levels=64, width=2048):
    """config 5's big-step construction (scripts/bench_stream.py: program_big) at nsteps steps of levels x width gates"""
    from scripts.bench_stream import make_steps
    nin = 256
    steps = make_steps(nsteps, levels, width, 0.25, nin)
    prim = list(range(nin))
    for k in range(1, nsteps):
        prim += [k * nin + i for i in range(steps[k - 1][0].num_outputs, nin)]
    return steps, prim


def _oracle_stream(steps, prim, key, rnd):
    og = oracle.Stream(key, rnd, prim)
    blocks = [bytes(og.garble(c.Gates, c.NumWires, in_, out_)) for c, in_, out_ in steps]
    return og, blocks


def _driver():
    from scripts import bench_stream
    if not os.path.exists(bench_stream.NATIVE):
        pytest.skip("tools/stream_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    return bench_stream


@pytest.mark.parametrize("mode", ["cooperative", "level-launches", "lost-workgroup"])
def test_c_host_big_steps_through_all_passes(mode):
    """8 steps of 131 072 gates (1.05e6 gates, 17.8 MB of stream), window 2, through the driver's six ways of handing bytes
    over — copying finish, deferred copies, views, evaluator per block, framed read buffers pageable and pinned; the driver checks
    the ways against one another (per-step checksums, the evaluated label against the garbler's pair), the test checks the
    stream's SHA-256 against the ORACLE's.  As cooperative launches, as one launch per level (GC_NO_COOP), and with the third
    cooperative pass losing a workgroup (repeated on the device, level launches from then on)."""
    bs = _driver()
    steps, prim = _big_program(8)
    assert all(c.NumGates == 131072 for c, _, _ in steps)
    key, rnd = bytes(range(32)), drbg("chost-big", 16 * (len(prim) + 1))
    _, blocks = _oracle_stream(steps, prim, key, rnd)
    want = hashlib.sha256(b"".join(blocks)).hexdigest()
    env = {"cooperative": {}, "level-launches": {"GC_NO_COOP": "1"}, "lost-workgroup": {"GC_COOP_FORCE_TIMEOUT": "3"}}[mode]
    r = bs.run_native_steps(steps, prim, rnd, key, 2, env=env)
    assert r["sha256"] == want and r["bytes"] == sum(len(b) for b in blocks) and r["steps"] == 8
    assert r["eval_blocks_matched"] + r["eval_blocks_parsed"] >= 8
    for k in ("garble_s", "garble_async_s", "garble_view_s", "eval_s", "eval_blocks_s", "eval_blocks_pinned_s"):
        assert r[k] > 0, k  # (every pass ran: the pinned one is skipped when gc_host_alloc refuses — said on stderr)
    if mode == "cooperative":
        assert r["coop_state"] in (1, -1)  # (-1: this context's self-test of the placement failed: level launches, same bytes)
        if r["coop_state"] == 1:
            assert r["coop_timeouts"] == 0
    elif mode == "level-launches":
        assert r["coop_state"] == -1 and r["coop_timeouts"] == 0
    elif r["coop_state"] != 0 and r["coop_timeouts"]:
        assert r["coop_state"] == -1 and r["coop_timeouts"] == 1


def test_c_host_ed25519like_one_digit_through_all_passes():
    """one scalar digit of the Ed25519-shaped program (2 605 steps, 1.04e7 gates: chains, fused units, the planner thread) with
    1 024 instructions in flight, all six passes; SHA-256 = the oracle-made golden"""
    bs = _driver()
    r = bs.run_native("ed25519like1", bytes(range(32)), 1024)
    assert r["sha256_ok"] is True and r["steps"] == 2605
    assert r["garble_view_gates_per_s"] and r["garble_async_gates_per_s"] and r["eval_blocks_pinned_gates_per_s"]


def test_c_host_mixed_program_through_all_passes():
    """grouped, deep and big steps in one program (the `mixed` mix at 600 instructions: 64-bit adders and multipliers, 131 072-gate
    steps in between), 64 in flight"""
    bs = _driver()
    steps, prim = bs.program_mixed(600)
    key, rnd = bytes(range(32)), drbg("chost-mixed", 16 * (len(prim) + 1))
    _, blocks = _oracle_stream(steps, prim, key, rnd)
    r = bs.run_native_steps(steps, prim, rnd, key, 64)
    assert r["sha256"] == hashlib.sha256(b"".join(blocks)).hexdigest()
    assert r["steps"] == len(steps) and max(c.NumGates for c, _, _ in steps) == 131072


def _framed(steps, blocks):
    out, starts = bytearray(), []
    for k, ((c, in_, out_), data) in enumerate(zip(steps, blocks)):
        starts.append(len(out))
        out += struct.pack(">5I", 1, k, c.NumGates, c.NumWires, max(max(in_), max(out_)) + 1) + data
    return bytes(out), starts


@pytest.mark.parametrize("no_coop", [False, True])
def test_python_host_big_steps_async_view_and_read_buffers(no_coop, monkeypatch):
    """gc_stream_garble_finish_async / _finish_view / _finish mixed over 6 steps of 131 072 gates (the three kinds may be mixed:
    include/gcengine.h), then the framed stream through gc_stream_eval_blocks from a pageable buffer in 1 MiB pieces and from a
    PINNED buffer (gc_host_alloc) in one piece: every byte and every evaluated output label against the oracle"""
    if no_coop:
        monkeypatch.setenv("GC_NO_COOP", "1")
    ctx = engine.Context(0)
    steps, prim = _big_program(6)
    key, rnd = drbg("pybig-key", 32), drbg("pybig", 16 * (len(prim) + 1))
    og, blocks = _oracle_stream(steps, prim, key, rnd)
    total = sum(len(b) for b in blocks)
    gg = engine.Stream(ctx, key, rnd, prim)
    handles = {}
    for c, in_, out_ in steps:
        if id(c) not in handles:
            handles[id(c)] = gg.intern(c.Gates, c.NumWires, len(in_), len(out_))
    dst = np.zeros(total + 64, np.uint8)
    issued, off, spans, direct = 0, 0, [], {}
    for k in range(len(steps)):
        while issued < min(len(steps), k + 2):
            c, in_, out_ = steps[issued]
            gg.garble_begin_h(handles[id(c)], in_, out_)
            issued += 1
        if k % 3 == 0:
            n = gg.garble_finish_async(dst, off)
            spans.append((k, off, n))
            off += n
        elif k % 3 == 1:
            direct[k] = gg.garble_finish_view()
        else:
            direct[k] = gg.garble_finish()
    gg.copies_wait()
    for k, o, n in spans:
        assert dst[o:o + n].tobytes() == blocks[k], "step %d (deferred copy)" % k
    for k, b in direct.items():
        assert b == blocks[k], "step %d" % k
    for o in steps[-1][2][::37]:
        assert gg.get(o) == og.get(o)
    # the evaluator: per block (the reference's loop), framed pageable, framed pinned
    framed, starts = _framed(steps, blocks)
    oe = oracle.StreamEval(key)
    evs = [engine.StreamEval(ctx, key) for _ in range(3)]
    bits = np.frombuffer(drbg("pybig-bits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = og.get(w)
        lab = wire["l1"] if b else wire["l0"]
        oe.set(w, lab)
        for ev in evs:
            ev.set(w, lab)
    for (c, in_, out_), data in zip(steps, blocks):
        nw = max(max(in_), max(out_)) + 1
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert evs[0].circuit(c.NumGates, c.NumWires, nw, data) == len(data)
    pos, done, win = 0, 0, 1 << 20
    while done < len(steps):
        used, nb, more = evs[1].blocks(framed[pos:pos + win])
        assert pos + used in starts + [len(framed)]
        win = win * 2 if used == 0 else 1 << 20
        assert used or more
        pos, done = pos + used, done + nb
    pin = engine.PinnedArray((len(framed),), np.uint8)
    pin.a[:] = np.frombuffer(framed, np.uint8)
    assert evs[2].blocks_at(pin.ptr, len(framed)) == (len(framed), len(steps), False)
    for k, (c, in_, out_) in enumerate(steps):
        for o in out_[::11]:
            want = oe.get(o)
            for i, ev in enumerate(evs):
                assert ev.get(o) == want, "step %d wire %d (evaluator %d)" % (k, o, i)
    ctx.sync()
    for ev in evs:
        ev.close()
    pin.close()
    gg.close()
    ctx.close()


def test_view_of_a_groups_last_step_with_deferred_copies_pending():
    """the steps of ONE group handed out by finish_async, its last step by finish_view — the slot stays with
    the copier threads until they are through; the next finish neither resets it under them nor leaks it.  Many rounds, so that
    slots are re-used while copies are still under way; bytes against the oracle."""
    from mpc_amd.circuit import adder
    ctx = engine.Context(0)
    c = adder(64)
    nin, nout = c.num_inputs, c.num_outputs
    rounds, per = 60, 24
    prim = list(range(nin * per))
    key, rnd = drbg("viewlast-key", 32), drbg("viewlast", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    h = gg.intern(c.Gates, c.NumWires, nin, nout)
    steps = []
    base = len(prim)
    for r in range(rounds):
        for j in range(per):  # independent steps: one group per round
            in_ = list(range(j * nin, (j + 1) * nin))
            out_ = list(range(base, base + nout))
            base += nout
            steps.append((in_, out_))
    want = [bytes(og.garble(c.Gates, c.NumWires, in_, out_)) for in_, out_ in steps]
    dst = np.zeros(sum(len(w) for w in want) + 64, np.uint8)
    off, spans = 0, []
    for r in range(rounds):
        for in_, out_ in steps[r * per:(r + 1) * per]:
            gg.garble_begin_h(h, in_, out_)
        for j in range(per):
            k = r * per + j
            if j == per - 1:
                assert gg.garble_finish_view() == want[k], "step %d (view of the group's last step)" % k
            else:
                n = gg.garble_finish_async(dst, off)
                spans.append((k, off, n))
                off += n
    gg.copies_wait()
    for k, o, n in spans:
        assert dst[o:o + n].tobytes() == want[k], "step %d (deferred copy)" % k
    gg.close()
    ctx.close()
