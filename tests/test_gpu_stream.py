"""GPU streaming garbler (gc_stream_*, config 5 shape): the byte stream must equal the oracle's
restatement of circuit/stream_garble.go byte for byte, for a sequence of per-step circuits."""
import os

import numpy as np
import pytest

import oracle
from mpc_amd import engine
from tests.test_oracle_stream import make_program
from tests.util import drbg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("base,keylen", [(0, 32), (0x20000, 16), (70000, 24)])
def test_stream_bytes_match_oracle(base, keylen):
    ctx = engine.Context(0)
    steps, prim = make_program(base)
    key = drbg("skey", keylen)
    rnd = drbg("gs%d" % base, 16 * (len(prim) + 1))
    og = oracle.Stream(key, rnd, prim)
    gg = engine.Stream(ctx, key, rnd, prim)
    for rep in range(2):  # second pass re-uses the cached plans
        for c, in_, out_ in steps:
            want = og.garble(c.Gates, c.NumWires, in_, out_)
            got = gg.garble(c.Gates, c.NumWires, in_, out_)
            assert got == want
            for o in out_:
                assert gg.get(o) == og.get(o)
    for w in prim:
        assert gg.get(w) == og.get(w)
    gg.close()
    ctx.close()


def test_stream_errors():
    ctx = engine.Context(0)
    steps, prim = make_program(0)
    with pytest.raises(engine.EngineError) as e:
        engine.Stream(ctx, bytes(5), drbg("r", 16 * (len(prim) + 1)), prim)
    assert e.value.code == engine.GC_E_KEYSIZE
    with pytest.raises(engine.EngineError) as e:
        engine.Stream(ctx, bytes(16), drbg("r", 16 * len(prim)), prim)
    assert e.value.code == engine.GC_E_RAND
    # evaluator: a block that reads a tmp wire before writing it is rejected (a tmp is private to its OpCircuit
    # block, stream_garble.go:131-157), and so is an unknown operation
    ge = engine.StreamEval(ctx, bytes(16))
    ge.set(0, (1, 2))
    with pytest.raises(engine.EngineError) as e:
        ge.circuit(1, 4, 4, bytes([0x90, 0, 0, 0, 0, 0, 1]))
    assert e.value.code == engine.GC_E_ARG
    with pytest.raises(engine.EngineError) as e:
        ge.circuit(1, 4, 4, bytes([0x10 | 9, 0, 0, 0, 0, 0, 1]))
    assert e.value.code == engine.GC_E_GATE
    # the block is the peer's data: more gates announced than the bytes can hold, or a global wire id beyond the
    # block's own numWires, are rejected before anything is sized by them (no std::bad_alloc across the C ABI)
    with pytest.raises(engine.EngineError) as e:
        ge.circuit(0x7fffffff, 4, 4, bytes([0x10, 0, 0, 0, 0, 0, 1]))
    assert e.value.code == engine.GC_E_ROWS
    with pytest.raises(engine.EngineError) as e:  # XOR 0, 0 -> global wire 0xffff of a 4-wire block
        ge.circuit(1, 4, 4, bytes([0x10, 0, 0, 0, 0, 0xff, 0xff]))
    assert e.value.code == engine.GC_E_ARG
    with pytest.raises(engine.EngineError) as e:  # long form, reads global wire 0xfffffff0
        ge.circuit(1, 4, 4, bytes([0x00, 0xff, 0xff, 0xff, 0xf0, 0, 0, 0, 0, 0, 0, 0, 1]))
    assert e.value.code == engine.GC_E_ARG
    ge.close()
    ctx.close()


def test_stream_in_out_alias_the_same_global_wire():
    """in[] and out[] naming the same global wire (in-place update): a gate reading that input AFTER the gate that set
    the output sees the new label, as the reference's per-gate stream.wire() look-up does (stream_garble.go:131-157)"""
    from mpc_amd.circuit import GATE
    ctx = engine.Context(0)
    key, prim = drbg("alias", 32), [0, 1, 2]
    rnd = drbg("alias-rnd", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    # circuit wires: 0, 1 inputs; 2 tmp; 3, 4 outputs.  global: in = [0, 1], out = [0, 2]  (wire 3 updates global 0)
    gates = np.zeros(4, GATE)
    gates[0] = (0, 1, 2, 2, 0)   # tmp = AND(w0, w1)       reads the OLD global 0
    gates[1] = (2, 1, 3, 0, 0)   # w3 (global 0) = tmp ^ w1
    gates[2] = (0, 1, 4, 2, 0)   # w4 (global 2) = AND(w0, w1): w0 now resolves to the NEW global 0
    gates[3] = (4, 0, 4, 0, 0)   # w4 ^= w0 (a re-written wire, new global 0 again)
    for rep in range(2):
        want = og.garble(gates, 5, [0, 1], [0, 2])
        got = gg.garble(gates, 5, [0, 1], [0, 2])
        assert got == want
        for w in (0, 1, 2):
            assert gg.get(w) == og.get(w)
    gg.close()
    ctx.close()


@pytest.mark.parametrize("base,keylen", [(0, 32), (0x20000, 16)])
def test_stream_evaluator_matches_oracle_and_plaintext(base, keylen):
    """garble on the GPU, evaluate the byte stream on the GPU (gc_stream_eval_*) and with the oracle's restatement
    of stream_evaluator.go; both must decode to the plaintext result"""
    from tests.test_oracle_stream import plain_program
    ctx = engine.Context(0)
    steps, prim = make_program(base)
    key = drbg("skey", keylen)
    rnd = drbg("ge%d" % base, 16 * (len(prim) + 1))
    gg = engine.Stream(ctx, key, rnd, prim)
    ge = engine.StreamEval(ctx, key)
    oe = oracle.StreamEval(key)
    bits = np.frombuffer(drbg("sbits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = gg.get(w)
        lab = wire["l1"] if b else wire["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    for c, in_, out_ in steps:
        data = gg.garble(c.Gates, c.NumWires, in_, out_)
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        for o in out_:
            assert ge.get(o) == oe.get(o)
    val = plain_program(steps, prim, bits)
    for c, in_, out_ in steps:
        for o in out_:
            wire = gg.get(o)
            want = wire["l1"] if val[o] else wire["l0"]
            assert ge.get(o) == (int(want["d0"]), int(want["d1"]))
    with pytest.raises(engine.EngineError) as e:  # truncated stream
        ge.circuit(steps[0][0].NumGates, steps[0][0].NumWires, 600, data[: len(data) // 2])
    assert e.value.code in (engine.GC_E_ROWS, engine.GC_E_GATE, engine.GC_E_ARG)
    gg.close(); ge.close(); ctx.close()


def test_stream_large_chained_program_matches_oracle():
    """the construction of scripts/bench_stream.py (config 5 shape: large per-step circuits chained through global
    wire ids) at a size the oracle restates in a second: byte streams equal, SHA-256 over the whole stream equal"""
    import hashlib
    from scripts.bench_stream import make_steps
    ctx = engine.Context(0)
    nin = 256
    steps = make_steps(5, 16, 1024, 0.25, nin)
    prim = list(range(nin))
    for k in range(1, len(steps)):
        prim += [k * nin + i for i in range(steps[k - 1][0].num_outputs, nin)]
    key = drbg("bigstream", 32)
    rnd = drbg("bigstream-rnd", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    h0, h1 = hashlib.sha256(), hashlib.sha256()
    for c, in_, out_ in steps:
        want = og.garble(c.Gates, c.NumWires, in_, out_)
        got = gg.garble(c.Gates, c.NumWires, in_, out_)
        assert len(got) == len(want) and got == want
        h0.update(want)
        h1.update(got)
    assert h0.digest() == h1.digest()
    assert sum(c.NumGates for c, _, _ in steps) == 5 * 16 * 1024
    gg.close()
    ctx.close()


def test_stream_big_steps_cooperative_and_level_launches_agree(monkeypatch):
    """a big step runs as ONE cooperative launch (32 workgroups of one XCD behind a barrier in its L2, k_garble_coop /
    k_eval_coop) when the context's self-test of that placement passes, and as one launch per level otherwise
    (GC_NO_COOP): the same bytes and the same evaluated labels either way, equal to the oracle's"""
    from scripts.bench_stream import make_steps
    nin = 256
    steps = make_steps(3, 24, 2048, 0.3, nin)
    prim = list(range(nin))
    for k in range(1, len(steps)):
        prim += [k * nin + i for i in range(steps[k - 1][0].num_outputs, nin)]
    key = drbg("coopstream", 32)
    rnd = drbg("coopstream-rnd", 16 * (len(prim) + 1))
    og = oracle.Stream(key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    outs = []
    for no_coop in (False, True):
        if no_coop:
            monkeypatch.setenv("GC_NO_COOP", "1")  # read when a context first meets a big step
        ctx = engine.Context(0)
        gg, ge = engine.Stream(ctx, key, rnd, prim), engine.StreamEval(ctx, key)
        for w in prim:
            ge.set(w, gg.get(w)["l0"])
        for (c, in_, out_), wbytes in zip(steps, want):
            got = gg.garble(c.Gates, c.NumWires, in_, out_)
            assert got == wbytes, "cooperative" if not no_coop else "level launches"
            nw = max(max(in_), max(out_)) + 1
            assert ge.circuit(c.NumGates, c.NumWires, nw, got) == len(got)
        outs.append([ge.get(o) for o in steps[-1][2]])
        ctx.sync()  # reports a cooperative pass that lost a workgroup
        gg.close(); ge.close(); ctx.close()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("no_coop", [False, True])
def test_stream_concurrent_cooperative_passes(no_coop, monkeypatch):
    """three garbler streams (three contexts, three threads) send big steps to the same GPU at once: their cooperative
    passes share the XCD's resident slots; every stream's bytes are the oracle's and no pass reports a lost workgroup.
    GC_NO_COOP: the level launches instead — launched directly, not recorded, while several contexts are alive (a launch on
    another thread's stream invalidates a stream capture under way: engine.cpp, live_contexts)"""
    if no_coop:
        monkeypatch.setenv("GC_NO_COOP", "1")
    import hashlib
    import threading
    from scripts.bench_stream import make_steps
    nin = 256
    steps = make_steps(12, 32, 2048, 0.25, nin)
    prim = list(range(nin))
    for k in range(1, len(steps)):
        prim += [k * nin + i for i in range(steps[k - 1][0].num_outputs, nin)]
    key = drbg("cc", 32)
    rnd = drbg("cc-rnd", 16 * (len(prim) + 1))
    og = oracle.Stream(key, rnd, prim)
    want = hashlib.sha256()
    for c, in_, out_ in steps:
        want.update(og.garble(c.Gates, c.NumWires, in_, out_))
    res = [None] * 3

    def run(i):
        try:
            ctx = engine.Context(0)
            gg = engine.Stream(ctx, key, rnd, prim)
            h = hashlib.sha256()
            for c, in_, out_ in steps:
                h.update(gg.garble(c.Gates, c.NumWires, in_, out_))
            ctx.sync()
            res[i] = h.hexdigest()
            gg.close(); ctx.close()
        except Exception as e:  # noqa: BLE001 - reported through the assertion below
            res[i] = "error: %s" % e

    th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert res == [want.hexdigest()] * 3


def test_stream_million_gate_step_matches_oracle():
    """one step of more than 2^20 gates: the device serialiser's block scan then runs more than one block per
    thread (k_ser_scan), and the evaluator renames > 10^6 wires; wire ids above 0xffff (long form) and below mix"""
    import hashlib
    from scripts.bench_stream import make_steps
    ctx = engine.Context(0)
    nin = 256
    steps = make_steps(1, 68, 16384, 0.2, nin)
    c, in_, out_ = steps[0]
    assert c.NumGates > (1 << 20)
    in_ = [0x10000 - 128 + i for i in range(nin)]  # half of the input ids take the long form
    prim = list(in_)
    key = drbg("mstream", 16)
    rnd = drbg("mstream-rnd", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    want = og.garble(c.Gates, c.NumWires, in_, out_)
    got = gg.garble(c.Gates, c.NumWires, in_, out_)
    assert len(got) == len(want)
    assert hashlib.sha256(got).digest() == hashlib.sha256(want).digest()
    # and the evaluator on that stream against the oracle's evaluator
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    for w in prim:
        lab = gg.get(w)["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    nw = max(max(in_), max(out_)) + 1
    assert ge.circuit(c.NumGates, c.NumWires, nw, got) == len(got)
    assert oe.circuit(c.NumGates, c.NumWires, nw, want) == len(want)
    for o in out_:
        assert ge.get(o) == oe.get(o)
    gg.close(); ge.close(); ctx.close()


def test_stream_matches_committed_golden(golden_dir):
    """the committed digests of tests/golden/stream_ot_golden.json (made with the oracle by make_golden_stream_ot.py)"""
    import importlib.util
    import json
    import os
    spec = importlib.util.spec_from_file_location("mk2", os.path.join(golden_dir, "make_golden_stream_ot.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = json.load(open(os.path.join(golden_dir, "stream_ot_golden.json")))
    ctx = engine.Context(0)
    for b, k in mk.STREAM_CASES:
        steps, prim, key, rnd = mk.stream_inputs(b, k)
        g = engine.Stream(ctx, key, rnd, prim)
        assert mk.stream_digest(g, steps) == gold["stream"]["%d/%d" % (b, k)]
        g.close()
    ctx.close()


def test_stream_outputs_that_are_input_wires():
    """a circuit whose last wires are input wires (in[] and out[] overlap): Get / Set resolve through in[] first
    (stream_garble.go:131-157), the overlapped output ids are never set; bytes equal to the oracle's"""
    from mpc_amd.circuit import AND, GATE, XOR, Circuit
    gates = np.zeros(2, GATE)
    gates[0] = (0, 1, 4, XOR, 0)
    gates[1] = (4, 2, 5, AND, 0)
    c = Circuit(6, [4], [3], gates)  # inputs 0..3, outputs = wires 3, 4, 5: wire 3 is an input
    in_, out_ = [10, 11, 12, 13], [20, 21, 22]
    key = drbg("ovk", 16)
    rnd = drbg("ovr", 16 * 5)
    ctx = engine.Context(0)
    og, gg = oracle.Stream(key, rnd, in_), engine.Stream(ctx, key, rnd, in_)
    assert gg.garble(c.Gates, c.NumWires, in_, out_) == og.garble(c.Gates, c.NumWires, in_, out_)
    for w in (21, 22):
        assert gg.get(w)["l0"] == og.get(w)["l0"]
    z = gg.get(20)["l0"]
    assert int(z["d0"]) == 0 and int(z["d1"]) == 0  # never set
    gg.close(); ctx.close()


def test_stream_pipelined_begin_finish_matches_oracle():
    """gc_stream_garble_begin / _finish with two circuits in flight: the bytes come out in order and equal the oracle's
    (so does every wire afterwards); the evaluator, whose calls return before the GPU is done, decodes the same stream to
    the oracle's labels"""
    ctx = engine.Context(0)
    steps, prim = make_program(0x20000)
    key = drbg("pkey", 32)
    rnd = drbg("pipe", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    bits = np.frombuffer(drbg("pbits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = gg.get(w)
        lab = wire["l1"] if b else wire["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    prog = steps * 3
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in prog]
    got = []
    gg.garble_begin(prog[0][0].Gates, prog[0][0].NumWires, prog[0][1], prog[0][2])
    for k in range(len(prog)):
        if k + 1 < len(prog):
            c1, in1, out1 = prog[k + 1]
            gg.garble_begin(c1.Gates, c1.NumWires, in1, out1)
        got.append(gg.garble_finish())
    assert got == want
    for c, in_, out_ in prog:
        for o in out_:
            assert gg.get(o) == og.get(o)
    for (c, in_, out_), data in zip(prog, got):  # evaluator: no waiting between blocks
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
    for c, in_, out_ in prog:
        for o in out_:
            assert ge.get(o) == oe.get(o)
    gg.close(); ge.close(); ctx.close()


def test_stream_evaluator_recognises_repeated_blocks():
    """one circuit streamed many times under different bindings: the evaluator decodes a block it has seen before only up
    to its rows and global ids (byte skeleton, gc_stream_eval_stats) — a binding with another repeat pattern among
    the ids, other id widths, or a changed byte must take the gate-by-gate path, and every label must equal the
    oracle's StreamEval either way"""
    if os.environ.get("GC_STREAM_NO_SKELETON"):
        pytest.skip("the byte skeletons are switched off (GC_STREAM_NO_SKELETON)")
    from mpc_amd.circuit import synthetic_levelised
    ctx = engine.Context(0)
    c = synthetic_levelised(6, 40, 0.4, seed=77, ninputs=12, inv_frac=0.1, xnor_frac=0.1)
    nout = c.num_outputs
    prim = list(range(40)) + [0x20000 + i for i in range(24)]
    key = drbg("skelkey", 32)
    rnd = drbg("skelrnd", 16 * (len(prim) + 1))
    gg = engine.Stream(ctx, key, rnd, prim)
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    os.environ["GC_STREAM_NO_SKELETON"] = "1"
    try:
        gp = engine.StreamEval(ctx, key)  # the same stream, every block parsed
    finally:
        del os.environ["GC_STREAM_NO_SKELETON"]
    bits = np.frombuffer(drbg("skelbits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = gg.get(w)
        lab = wire["l1"] if b else wire["l0"]
        for ev in (ge, oe, gp):
            ev.set(w, lab)
    outs = lambda base: [base + i for i in range(nout)]
    steps = [
        (prim[0:12], outs(1000), "parse"),                         # first sight
        (prim[12:24], outs(1100), "match"),                        # same pattern, other wires
        ([prim[0]] * 2 + prim[2:12], outs(1200), "parse"),         # two inputs name one wire: another circuit
        ([prim[5]] * 2 + prim[26:36], outs(1300), "match"),
        (outs(1000)[:12], [outs(1000)[0]] + outs(1400)[1:], "parse"),  # an output lands on an input wire
        (outs(1100)[:12], [outs(1100)[0]] + outs(1500)[1:], "match"),
        (prim[40:52], outs(0x30000), "parse"),                     # 32-bit ids: another byte layout
        (prim[52:64], outs(0x30100), "match"),
        (prim[24:36], outs(1600), "match"),                        # back to the first layout
    ]
    datas = []
    for in_, out_, how in steps:
        data = gg.garble(c.Gates, c.NumWires, in_, out_)
        datas.append((data, max(max(in_), max(out_)) + 1))
        nw = datas[-1][1]
        before = ge.stats()
        for ev in (ge, oe, gp):
            assert ev.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        after = ge.stats()
        assert (after[0] - before[0], after[1] - before[1]) == ((1, 0) if how == "parse" else (0, 1)), how
        for o in out_:
            assert ge.get(o) == oe.get(o) == gp.get(o)
    assert gp.stats()[1] == 0
    # a changed byte outside rows and global ids (an XOR's op byte made XNOR: same lengths) is not the skeleton's block
    data, nw = datas[1]
    xor_at = None
    pos = 0
    raw = bytearray(data)
    for g in range(c.NumGates):  # walk the records (stream_garble.go:391-446): op byte, ids, rows
        op = raw[pos] & 0x0f
        z = 2 if raw[pos] & 0x10 else 4
        if op == 0 and xor_at is None:
            xor_at = pos
        pos += 1 + z * (2 if op == 4 else 3) + 16 * {0: 0, 1: 0, 2: 2, 3: 3, 4: 1}[op]
    assert pos == len(raw) and xor_at is not None
    raw[xor_at] |= 1  # XOR -> XNOR
    before = ge.stats()
    for ev in (ge, oe):
        assert ev.circuit(c.NumGates, c.NumWires, nw, bytes(raw)) == len(raw)
    assert ge.stats()[0] == before[0] + 1
    for o in steps[1][1]:
        assert ge.get(o) == oe.get(o)
    gg.close(); ge.close(); gp.close(); ctx.close()


@pytest.mark.parametrize("threads", ["3", "0"])
def test_stream_evaluator_big_repeated_blocks_on_helper_threads(threads, monkeypatch):
    """big blocks that match a byte skeleton are compared / copied in eight segments by the caller and helper threads
    (SkelPool; GC_STREAM_THREADS = 0: the caller alone): four 49 152-gate blocks of one shape against the oracle's StreamEval,
    then a block with one byte changed in its LAST segment (an XOR made XNOR) — it is not the skeleton's block and has to
    take the gate-by-gate path"""
    from scripts.bench_stream import make_steps
    monkeypatch.setenv("GC_STREAM_THREADS", threads)
    ctx = engine.Context(0)
    nin = 256
    steps = make_steps(4, 24, 2048, 0.4, nin)
    steps = [(steps[0][0], i, o) for _, i, o in steps]  # one shape: blocks 2 .. 4 match block 1's skeleton
    prim = list(range(nin))
    for k in range(1, len(steps)):
        prim += [k * nin + i for i in range(steps[k - 1][0].num_outputs, nin)]
    key = drbg("bigskel", 32)
    rnd = drbg("bigskel-rnd", 16 * (len(prim) + 1))
    gg = engine.Stream(ctx, key, rnd, prim)
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    for w in prim:
        lab = gg.get(w)["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    datas = []
    for c, in_, out_ in steps:
        data = gg.garble(c.Gates, c.NumWires, in_, out_)
        nw = max(max(in_), max(out_)) + 1
        datas.append((c, data, nw, out_))
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        for o in out_[:16] + out_[-16:]:
            assert ge.get(o) == oe.get(o)
    assert ge.stats() == (1, 3)  # parsed, matched
    c, data, nw, out_ = datas[-1]
    raw = bytearray(data)
    pos, last_xor = 0, None
    for g in range(c.NumGates):  # walk the records (stream_garble.go:391-446): op byte, ids, rows
        op = raw[pos] & 0x0f
        z = 2 if raw[pos] & 0x10 else 4
        if op == 0:
            last_xor = pos
        pos += 1 + z * (2 if op == 4 else 3) + 16 * {0: 0, 1: 0, 2: 2, 3: 3, 4: 1}[op]
    assert pos == len(raw) and last_xor is not None and last_xor > len(raw) * 7 // 8
    raw[last_xor] |= 1  # XOR -> XNOR
    for ev in (ge, oe):
        assert ev.circuit(c.NumGates, c.NumWires, nw, bytes(raw)) == len(raw)
    assert ge.stats() == (2, 3)
    for o in out_:
        assert ge.get(o) == oe.get(o)
    gg.close(); ge.close(); ctx.close()


def _dependency_program(base):
    """A program of small SSA-step circuits with every kind of dependency between neighbours (global wires through
    in[] / out[]): runs of independent steps, read-after-write chains, two steps writing the same wire, a step
    overwriting a wire an earlier queued step reads, a big step in the middle, an in-place update."""
    from mpc_amd.circuit import comparator64, synthetic_levelised
    shapes = [
        synthetic_levelised(6, 40, 0.3, seed=51, ninputs=24, inv_frac=0.1, xnor_frac=0.1),
        synthetic_levelised(3, 170, 0.5, seed=52, ninputs=24, or_frac=0.1, inv_frac=0.05),   # OR gates: the HAS_OR build
        synthetic_levelised(12, 16, 0.25, seed=53, ninputs=24),
        comparator64(),
        synthetic_levelised(20, 200, 0.2, seed=54, ninputs=24, inv_frac=0.05),                # 4 000 gates
    ]
    big = synthetic_levelised(20, 2000, 0.2, seed=55, ninputs=24, inv_frac=0.05)              # 40 000 gates: its own sequence
    prim = [base + i for i in range(256)]
    nxt = [base + 1000]

    def fresh(n):
        w = list(range(nxt[0], nxt[0] + n))
        nxt[0] += n
        return w

    steps = []

    def add(c, in_, out_=None):
        out_ = fresh(c.num_outputs) if out_ is None else out_
        steps.append((c, in_, out_))
        return out_

    def ins(c, src, k):  # c.num_inputs wires from src, rotated by k
        n = c.num_inputs
        return [src[(k + i) % len(src)] for i in range(n)]

    # 1. sixteen independent steps (disjoint outputs, shared read-only inputs): one group
    outs = [add(shapes[k % 3], ins(shapes[k % 3], prim, 7 * k)) for k in range(16)]
    # 2. a read-after-write chain through them
    prev = outs[0] + outs[1]
    for k in range(5):
        c = shapes[(k + 1) % 3]
        prev = add(c, ins(c, prev + prim, k)) + prev
    # 3. two steps writing the SAME wires (the later one must win), then a reader
    tgt = fresh(shapes[0].num_outputs)
    add(shapes[0], ins(shapes[0], prim, 3), tgt)
    add(shapes[0], ins(shapes[0], prim, 90), tgt)
    add(shapes[2], ins(shapes[2], tgt + prim, 0))
    # 4. write-after-read: a queued step reads wires that the next step overwrites
    a = add(shapes[2], ins(shapes[2], prim, 11))
    add(shapes[0], ins(shapes[0], a + prim, 0))          # reads a
    add(shapes[2], ins(shapes[2], prim, 40), a)          # overwrites a: the reader before must have seen the old labels
    add(shapes[0], ins(shapes[0], a + prim, 0))          # reads the new a
    # 5. the comparator (128 inputs) and the 4 000-gate shape among independent small steps
    add(shapes[3], prim[:128])
    add(shapes[4], ins(shapes[4], prim, 17))
    for k in range(6):
        add(shapes[k % 3], ins(shapes[k % 3], prim, 5 * k + 1))
    # 6. a big step (more than 32 768 gates) between small ones, consuming and feeding them
    bo = add(big, ins(big, steps[-1][2] + prim, 0))
    add(shapes[0], ins(shapes[0], bo + prim, 2))
    # 7. in-place update: out[] names wires of in[]
    c = shapes[2]
    i7 = ins(c, prim, 60)
    add(c, i7, i7[: c.num_outputs])
    add(c, ins(c, i7 + prim, 1))
    # 8. a long run of independent steps again (more than one group's worth of dependencies must not accumulate)
    for k in range(40):
        add(shapes[k % 3], ins(shapes[k % 3], prim, 3 * k))
    return steps, prim


@pytest.mark.parametrize("base,keylen,window", [(0, 32, 64), (0x20000, 16, 7), (0xfe00, 24, 300)])
def test_stream_step_groups_match_oracle(base, keylen, window):
    """step-level parallelism: `window` circuits queued ahead (gc_stream_garble_begin) — independent small ones run side
    by side in one launch sequence, dependent ones in later groups — and every byte, in program order, equals the
    oracle's serial Streaming.Garble; so does every wire afterwards; the evaluator, fed all blocks before the first
    read-back, ends on the oracle's labels"""
    ctx = engine.Context(0)
    steps, prim = _dependency_program(base)
    key = drbg("gkey", keylen)
    rnd = drbg("grp%d" % base, 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    got = []
    issued = 0
    handles = {}
    for k in range(len(steps)):
        while issued < len(steps) and issued < k + window:
            c, in_, out_ = steps[issued]
            if window == 7:  # by content on every call
                gg.garble_begin(c.Gates, c.NumWires, in_, out_)
            else:            # by handle (gc_stream_intern once per circuit; also for the in-place update of step 7)
                if id(c) not in handles:
                    handles[id(c)] = gg.intern(c.Gates, c.NumWires, len(in_), len(out_))
                    assert gg.intern(c.Gates, c.NumWires, len(in_), len(out_)) == handles[id(c)]
                gg.garble_begin_h(handles[id(c)], in_, out_)
            issued += 1
        got.append(gg.garble_finish())
    for k, (w, g) in enumerate(zip(want, got)):
        assert g == w, "step %d of %d" % (k, len(steps))
    groups, grouped, bigs = gg.stats()
    assert bigs == 1 and grouped == len(steps) - 1
    if window >= 64:
        assert groups < grouped // 2, (groups, grouped)  # the independent runs really shared launches
    for c, in_, out_ in steps:
        for o in out_:
            assert gg.get(o) == og.get(o)
    # evaluator: every block handed over before anything is read back
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    bits = np.frombuffer(drbg("gbits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = og.get(w)
        lab = wire["l1"] if b else wire["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    for (c, in_, out_), data in zip(steps, want):
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
    for c, in_, out_ in steps:
        for o in out_:
            assert ge.get(o) == oe.get(o)
    gg.close(); ge.close(); ctx.close()


def test_stream_single_calls_still_match_after_groups():
    """gc_stream_garble (begin + finish, nothing in flight) on small circuits = groups of one; mixing it with queued
    windows and with get_wire in between keeps the bytes and the wires the oracle's"""
    ctx = engine.Context(0)
    steps, prim = _dependency_program(0)
    steps = steps[:30]
    key = drbg("gkey1", 32)
    rnd = drbg("grp1", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    for k, (c, in_, out_) in enumerate(steps):
        want = og.garble(c.Gates, c.NumWires, in_, out_)
        if k % 3 == 0:
            assert gg.garble(c.Gates, c.NumWires, in_, out_) == want
        else:
            gg.garble_begin(c.Gates, c.NumWires, in_, out_)
            if k % 3 == 1:
                gg.flush()
            assert gg.get(out_[0]) == og.get(out_[0])  # launches what is queued and waits
            assert gg.garble_finish() == want
    gg.close(); ctx.close()


def test_stream_circuit_cache_is_bounded(monkeypatch):
    """the per-stream circuit caches are capped (GC_STREAM_CACHE_GATES) with LRU eviction — a stream of ever
    new circuits keeps working and keeps matching the oracle on both sides when old circuits (and, on the evaluator's
    side, the byte skeletons that point at them) are dropped and later come back"""
    from mpc_amd.circuit import synthetic_levelised
    monkeypatch.setenv("GC_STREAM_CACHE_GATES", "1500")  # room for two or three of the circuits below
    ctx = engine.Context(0)
    circs = [synthetic_levelised(5, 100, 0.3, seed=300 + i, ninputs=16, inv_frac=0.1) for i in range(6)]
    prim = list(range(64))
    key = drbg("ckey", 32)
    rnd = drbg("crnd", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    for w in prim:
        lab = og.get(w)["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    base = 1000
    order = [0, 1, 2, 3, 4, 5, 0, 1, 0, 5, 2, 2, 3, 0]
    for n, ci in enumerate(order):
        c = circs[ci]
        in_ = [prim[(3 * n + i) % len(prim)] for i in range(16)]
        out_ = [base + 200 * n + i for i in range(c.num_outputs)]
        want = og.garble(c.Gates, c.NumWires, in_, out_)
        assert gg.garble(c.Gates, c.NumWires, in_, out_) == want
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, want) == len(want)
        assert oe.circuit(c.NumGates, c.NumWires, nw, want) == len(want)
        assert ge.get(out_[-1]) == oe.get(out_[-1])
    gg.close(); ge.close(); ctx.close()


def test_stream_big_steps_in_flight_while_their_circuits_are_evicted(monkeypatch):
    """big steps keep two table buffers in flight (the batch of the last step is held while the next one's pass runs; the
    serialiser and the evaluator's uploads run on their own streams) — with a cache that holds ONE of the four circuits,
    every step evicts the circuit whose batch is held: three steps queued ahead on the garbler, the evaluator fed the
    blocks back to back; bytes, and the labels of the last step, equal the oracle's"""
    from scripts.bench_stream import make_steps
    monkeypatch.setenv("GC_STREAM_CACHE_GATES", "60000")  # one 49 152-gate circuit
    nin = 256
    steps = make_steps(11, 24, 2048, 0.3, nin)
    prim = list(range(nin))
    for k in range(1, len(steps)):
        prim += [k * nin + i for i in range(steps[k - 1][0].num_outputs, nin)]
    key = drbg("evict-big", 32)
    rnd = drbg("evict-big-rnd", 16 * (len(prim) + 1))
    og, oe = oracle.Stream(key, rnd, prim), oracle.StreamEval(key)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    ctx = engine.Context(0)
    gg, ge = engine.Stream(ctx, key, rnd, prim), engine.StreamEval(ctx, key)
    for w in prim:
        lab = og.get(w)["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    got, issued = [], 0
    for k in range(len(steps)):
        while issued < min(len(steps), k + 3):
            c, in_, out_ = steps[issued]
            gg.garble_begin(c.Gates, c.NumWires, in_, out_)
            issued += 1
        got.append(gg.garble_finish())
    assert got == want
    for (c, in_, out_), wbytes in zip(steps, want):
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, wbytes) == len(wbytes)
        assert oe.circuit(c.NumGates, c.NumWires, nw, wbytes) == len(wbytes)
    assert [ge.get(o) for o in steps[-1][2]] == [oe.get(o) for o in steps[-1][2]]
    assert ge.stats()[0] == len(steps)  # every block was a first-time block again: nothing was kept
    ctx.sync()
    gg.close(); ge.close(); ctx.close()


@pytest.mark.parametrize("deps", [False, True])
def test_stream_mixed_program_matches_oracle(deps, monkeypatch):
    """(deps: with GC_STREAM_DEPS=1, the experiment of units that wait inside a launch — include/gcengine.h:
    gc_stream_wait_stats)  The mixed program of scripts/bench_stream.py (64-bit adders, 64 x 64 multipliers — 13 740 gates, still one
    workgroup each — and 131 072-gate steps, operands mostly from the last few steps, variables overwritten) at a size
    the oracle restates in a second: every step's bytes and, on the evaluator's side, every label equal the oracle's"""
    from scripts.bench_stream import program_mixed
    if deps:
        monkeypatch.setenv("GC_STREAM_DEPS", "1")
    ctx = engine.Context(0)
    steps, prim = program_mixed(160, seed=9, big_every=53)
    key = drbg("mixkey", 32)
    rnd = drbg("mixrnd", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    issued = 0
    for k in range(len(steps)):
        while issued < len(steps) and issued < k + 48:
            c, in_, out_ = steps[issued]
            gg.garble_begin(c.Gates, c.NumWires, in_, out_)
            issued += 1
        assert gg.garble_finish() == want[k], "step %d (%s)" % (k, steps[k][0].name)
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    bits = np.frombuffer(drbg("mixbits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = og.get(w)
        lab = wire["l1"] if b else wire["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    for k, ((c, in_, out_), data) in enumerate(zip(steps, want)):
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        if k % 37 == 36:  # read-backs in between launch what is queued
            assert ge.get(out_[0]) == oe.get(out_[0]), "step %d (%s)" % (k, c.name)
    for k, (c, in_, out_) in enumerate(steps):
        for o in out_[::7]:
            assert ge.get(o) == oe.get(o), "step %d (%s) wire %d" % (k, c.name, o)
    gg.close(); ge.close(); ctx.close()


@pytest.mark.parametrize("deps", [False, True])
def test_stream_instruction_mix_matches_oracle(deps, monkeypatch):
    """(deps: with GC_STREAM_DEPS=1, as above)  The 23-circuit instruction mix of scripts/bench_stream.py (ssa23: adders to 512 bits, array multipliers to 256 bits —
    those of 128 and 256 bits run on the planner's late schedule —, a comparator with OR gates, table-free blocks, mux /
    shift / hash-like blocks) at a size the oracle restates in seconds, every one of the 23 circuits at least once: every
    step's bytes and the evaluated labels equal the oracle's, evaluated on random input bits"""
    from scripts.bench_stream import program_ssa, ssa_circuits
    if deps:
        monkeypatch.setenv("GC_STREAM_DEPS", "1")
    ctx = engine.Context(0)
    steps, prim = program_ssa(220, seed=3)
    seen = {id(c) for c, _, _ in steps}
    pools = {}
    for c, in_, _ in steps:
        pools.setdefault(c.num_inputs // 2, in_[:c.num_inputs // 2])
    nxt = max(max(max(i), max(o)) for _, i, o in steps) + 1
    for _, c, _ in ssa_circuits():  # (the cached circuit objects differ: compare by shape)
        if not any(s[0].name == c.name and s[0].NumGates == c.NumGates for s in steps):
            w = c.num_inputs // 2
            base = pools.get(w)
            if base is None:
                base = list(range(nxt, nxt + w)); prim = prim + base; nxt += w
                pools[w] = base
            steps.append((c, base + base, list(range(nxt, nxt + c.num_outputs))))
            nxt += c.num_outputs
    assert len({(c.name, c.NumGates) for c, _, _ in steps}) == 23 and seen
    key = drbg("ssakey", 32)
    rnd = drbg("ssarnd", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    issued = 0
    for k in range(len(steps)):
        while issued < len(steps) and issued < k + 32:
            c, in_, out_ = steps[issued]
            gg.garble_begin(c.Gates, c.NumWires, in_, out_)
            issued += 1
        assert gg.garble_finish() == want[k], "step %d (%s)" % (k, steps[k][0].name)
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    bits = np.frombuffer(drbg("ssabits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = og.get(w)
        lab = wire["l1"] if b else wire["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    for (c, in_, out_), data in zip(steps, want):
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
    for k, (c, in_, out_) in enumerate(steps):
        for o in out_[::5]:
            assert ge.get(o) == oe.get(o), "step %d (%s) wire %d" % (k, c.name, o)
    ctx.sync()
    gg.close(); ge.close(); ctx.close()


def test_stream_closed_with_big_steps_in_flight():
    """a garbler closed with big steps begun and not finished, an evaluator closed right behind its last big block (uploads on
    their own stream, a batch held for the next step, the serialiser still running): nothing is waited for by the caller,
    nothing may be freed under a kernel — and the context is as good as new afterwards (a second stream on it matches the
    oracle)"""
    from scripts.bench_stream import make_steps
    nin = 256
    steps = make_steps(5, 24, 2048, 0.3, nin)
    prim = list(range(nin))
    for k in range(1, len(steps)):
        prim += [k * nin + i for i in range(steps[k - 1][0].num_outputs, nin)]
    key = drbg("close-big", 32)
    rnd = drbg("close-big-rnd", 16 * (len(prim) + 1))
    og = oracle.Stream(key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    ctx = engine.Context(0)
    for rounds in range(3):
        gg = engine.Stream(ctx, key, rnd, prim)
        for c, in_, out_ in steps[:3]:
            gg.garble_begin(c.Gates, c.NumWires, in_, out_)
        assert gg.garble_finish() == want[0]
        gg.close()  # two steps still in flight
        ge = engine.StreamEval(ctx, key)
        for w in prim:
            ge.set(w, og.get(w)["l0"])
        for (c, in_, out_), data in zip(steps[:3], want):
            nw = max(max(in_), max(out_)) + 1
            assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        ge.close()  # the last passes may still be running
    gg = engine.Stream(ctx, key, rnd, prim)
    assert [gg.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps] == want
    ctx.sync()
    gg.close(); ctx.close()


# ---- deep lanes: long one-workgroup steps on streams of their own, beside the step groups (round 4) ------------------------

def _deep_dependency_program(base):
    """Every kind of dependency ACROSS the two scheduling classes.  With GC_STREAM_DEEP_STEPS = 100 the 128- and 256-bit
    adders / subtractors (255 / 511 barriers) are DEEP steps (a lane each), the 64-bit ones and the bitwise blocks are small
    (step groups on the ctx stream); so is a 64-bit array multiplier (179 barriers); a 40 000-gate wide circuit is a BIG step (a
    pass of the ctx stream)."""
    from mpc_amd.circuit import AND, XOR, adder, bitwise, multiplier, subtractor, synthetic_levelised
    add64, add128, add256 = adder(64), adder(128), adder(256)
    sub128 = subtractor(128)
    xor128, and128, xor64 = bitwise(128, XOR), bitwise(128, AND), bitwise(64, XOR)
    mul64 = multiplier(64)
    big = synthetic_levelised(20, 2000, 0.2, seed=55, ninputs=256, inv_frac=0.05)
    prim = [base + i for i in range(1024)]
    v = lambda k, n: prim[(k * 64) % 768:(k * 64) % 768 + n]  # an input value of n bits
    nxt = [base + 2000]
    steps = []

    def add(c, a, b, out_=None):
        if out_ is None:
            out_ = list(range(nxt[0], nxt[0] + c.num_outputs))
            nxt[0] += c.num_outputs
        steps.append((c, a + b, out_))
        return out_

    # 1. independent deep steps (they spread over the lanes) between independent small ones
    d = [add(add128, v(k, 128), v(k + 2, 128)) for k in range(4)]
    s = [add(xor64, v(k, 64), v(k + 1, 64)) for k in range(6)]
    # 2. read after write: small <- deep, deep <- small, deep <- deep (other lane), small <- small <- deep
    x = add(xor128, d[0], d[1])                       # a group waits for two deep steps
    y = add(sub128, x, d[2])                          # a deep step waits for a group and for a deep step
    z = add(add128, y, d[3])                          # deep <- deep
    w = add(and128, z, x)
    w = add(xor128, w, y)
    # 3. write after write: two deep steps, then a small and a deep step, write the SAME wires; the reader sees the last
    tgt = list(range(nxt[0], nxt[0] + 128)); nxt[0] += 128
    add(add128, v(1, 128), v(3, 128), tgt)
    add(sub128, v(5, 128), v(2, 128), tgt)
    r1 = add(xor128, tgt, v(4, 128))
    add(xor128, v(6, 128), v(7, 128), tgt)            # small over deep
    add(add128, v(8, 128), v(9, 128), tgt)            # deep over small
    r2 = add(and128, tgt, r1)
    # 4. write after read: a deep step reads wires that a small step overwrites right after, and the other way round
    a = add(xor128, v(2, 128), v(6, 128))
    r3 = add(add128, a, v(1, 128))                    # deep reads a
    add(and128, v(3, 128), v(5, 128), a)              # small overwrites a: the deep step must have read the old labels
    r4 = add(xor128, a, r3)
    b = add(xor128, v(7, 128), v(0, 128))
    r5 = add(and128, b, v(2, 128))                    # small reads b
    add(sub128, v(4, 128), v(8, 128), b)              # deep overwrites b
    r6 = add(xor128, b, r5)
    # 5. several deep readers of one value on different lanes, then a writer of that value (it must wait for all of them)
    c0 = add(xor128, v(1, 128), v(9, 128))
    rs = [add(add128, c0, v(k, 128)) for k in range(4)]
    add(xor128, v(3, 128), v(4, 128), c0)
    r7 = add(add128, c0, rs[3])
    # 6. in-place updates: out[] names wires of in[] — a deep step, then a small one on the same value
    acc = add(xor128, v(0, 128), v(5, 128))
    add(add128, acc, v(2, 128), acc)
    add(xor128, acc, v(3, 128), acc)
    add(sub128, acc, v(7, 128), acc)
    # 7. a big step (a pass of the ctx stream) between deep ones, consuming and feeding them; a long small step beside
    p256 = add(add256, v(0, 256), v(4, 256))
    bo = add(big, p256[:128], r7)
    q = add(add128, bo[:128], bo[:128])
    m = add(mul64, q[:64], s[0])
    add(xor64, m, s[1])
    # 8. a chain of deep steps (one lane's worth of order) while small steps stream by
    ch = d[0]
    for k in range(6):
        ch = add(add128 if k % 2 else sub128, ch, v(k, 128))
        for j in range(5):
            add(xor64, v(k + j, 64), s[j])
    add(xor128, ch, r6)
    add(xor128, r2, r4)
    return steps, prim


@pytest.mark.parametrize("base,keylen,lanes,by_handle", [(0, 32, None, True), (0x20000, 16, "1", False), (0xfe00, 24, "2", True),
                                                         (0, 32, "0", True), (0x20000, 32, "nofollow", False)])
def test_stream_deep_lanes_match_oracle(base, keylen, lanes, by_handle, monkeypatch):
    """deep steps (lanes) beside step groups and a big step: every byte in program order, every wire afterwards and the
    evaluator's labels equal the oracle's serial loop, with 3 / 1 / 2 lanes, with the lanes switched off, and with short
    steps kept off the lanes (GC_STREAM_NO_FOLLOW: they wait for the deep steps they depend on in a group instead)"""
    monkeypatch.setenv("GC_STREAM_DEEP_STEPS", "100")
    if lanes == "nofollow":
        monkeypatch.setenv("GC_STREAM_NO_FOLLOW", "1")
        lanes = None
        nofollow = True
    else:
        nofollow = False
    if lanes is not None:
        monkeypatch.setenv("GC_STREAM_DEEP_LANES", lanes)
    ctx = engine.Context(0)
    steps, prim = _deep_dependency_program(base)
    key = drbg("deepkey", keylen)
    rnd = drbg("deep%d" % base, 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    got, issued, handles = [], 0, {}
    for k in range(len(steps)):
        while issued < len(steps) and issued < k + 48:
            c, in_, out_ = steps[issued]
            if by_handle:
                if id(c) not in handles:
                    handles[id(c)] = gg.intern(c.Gates, c.NumWires, len(in_), len(out_))
                gg.garble_begin_h(handles[id(c)], in_, out_)
            else:
                gg.garble_begin(c.Gates, c.NumWires, in_, out_)
            issued += 1
        got.append(gg.garble_finish())
    for k, (w, g) in enumerate(zip(want, got)):
        assert g == w, "step %d of %d (%s)" % (k, len(steps), steps[k][0].name)
    deep_steps, nlanes = gg.deep_stats()
    # (with the threshold at 100 barriers the 64-bit multiplier — 179 — is a deep step as well)
    ndeep = sum(1 for c, _, _ in steps if c.name in ("adder128", "subtractor128", "adder256", "multiplier64"))
    if lanes == "0" or (lanes is None and os.environ.get("GC_STREAM_DEEP_LANES") == "0"):  # (also: the suite run with lanes off)
        assert (deep_steps, nlanes) == (0, 0)
        lanes = "0"
    elif nofollow:
        assert nlanes >= 1 and deep_steps == ndeep, (deep_steps, ndeep, nlanes)
    else:  # (short steps that depend on a deep step in flight follow it onto its lane: they count as well)
        assert nlanes >= 1 and deep_steps >= ndeep, (deep_steps, ndeep, nlanes)
    for c, in_, out_ in steps:
        for o in out_[::7]:
            assert gg.get(o) == og.get(o)
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    bits = np.frombuffer(drbg("deepbits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = og.get(w)
        lab = wire["l1"] if b else wire["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    for (c, in_, out_), data in zip(steps, want):
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
    if lanes != "0":
        assert ge.deep_stats()[0] >= ndeep
    for k, (c, in_, out_) in enumerate(steps):
        for o in out_[::7]:
            assert ge.get(o) == oe.get(o), "step %d (%s) wire %d" % (k, c.name, o)
    ctx.sync()
    gg.close(); ge.close(); ctx.close()


def test_stream_deep_steps_in_flight_while_circuits_are_evicted(monkeypatch):
    """a circuit cache too small for the program's circuits, deep steps in flight on their lanes when room is made: the lanes
    are drained before a circuit goes, the bytes stay the oracle's (both sides)"""
    from mpc_amd.circuit import AND, XOR, adder, bitwise, subtractor
    monkeypatch.setenv("GC_STREAM_DEEP_STEPS", "100")
    monkeypatch.setenv("GC_STREAM_CACHE_GATES", "2500")
    shapes = [adder(128), subtractor(128), bitwise(128, XOR), adder(256), bitwise(128, AND), adder(192), subtractor(160)]
    prim = list(range(512))
    steps, nxt = [], 600
    prev = prim[:128]
    for k in range(40):
        c = shapes[k % len(shapes)]
        w = c.num_inputs // 2
        a = (prev + prim)[:w]
        b = prim[(k * 32) % 256:(k * 32) % 256 + w]
        out_ = list(range(nxt, nxt + c.num_outputs))
        nxt += c.num_outputs
        steps.append((c, a + b, out_))
        if k % 3 == 0:
            prev = out_
    key = drbg("deep-evict", 32)
    rnd = drbg("deep-evict-rnd", 16 * (len(prim) + 1))
    ctx = engine.Context(0)
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    issued = 0
    for k in range(len(steps)):
        while issued < len(steps) and issued < k + 12:
            c, in_, out_ = steps[issued]
            gg.garble_begin(c.Gates, c.NumWires, in_, out_)
            issued += 1
        assert gg.garble_finish() == want[k], "step %d (%s)" % (k, steps[k][0].name)
    if os.environ.get("GC_STREAM_DEEP_LANES") != "0":  # (the suite also runs with the lanes switched off)
        assert gg.deep_stats()[0] > 10
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    for w in prim:
        ge.set(w, og.get(w)["l0"])
        oe.set(w, og.get(w)["l0"])
    for (c, in_, out_), data in zip(steps, want):
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
    for c, in_, out_ in steps:
        for o in out_[::9]:
            assert ge.get(o) == oe.get(o)
    ctx.sync()
    gg.close(); ge.close(); ctx.close()


def test_stream_ed25519like_matches_oracle():
    """one scalar digit of the Ed25519-shaped program of scripts/bench_stream.py (program_ed25519: selectPoint, geMixedAdd,
    ToExtended of the reference's ed25519.mpcl, instruction by instruction — 2 605 steps, 1.04e7 gates, constants as wires,
    operands with repeated sign wires): every step's bytes and the evaluator's labels equal the oracle's"""
    from scripts.bench_stream import PROGRAMS
    steps, prim = PROGRAMS["ed25519like1"]()
    key = drbg("edkey", 32)
    rnd = drbg("edrnd", 16 * (len(prim) + 1))
    ctx = engine.Context(0)
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    handles, issued = {}, 0
    for k in range(len(steps)):
        while issued < len(steps) and issued < k + 64:
            c, in_, out_ = steps[issued]
            if id(c) not in handles:
                handles[id(c)] = gg.intern(c.Gates, c.NumWires, len(in_), len(out_))
            gg.garble_begin_h(handles[id(c)], in_, out_)
            issued += 1
        assert gg.garble_finish() == want[k], "step %d (%s)" % (k, steps[k][0].name)
    groups, grouped, bigs = gg.stats()
    assert bigs == 0
    if not os.environ.get("GC_STREAM_OPEN_GROUPS"):  # (the default window: the hundred products of a FeMul share launches)
        assert groups < len(steps) // 4, (groups, grouped, bigs)
    ge, oe = engine.StreamEval(ctx, key), oracle.StreamEval(key)
    bits = np.frombuffer(drbg("edbits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = og.get(w)
        lab = wire["l1"] if b else wire["l0"]
        ge.set(w, lab)
        oe.set(w, lab)
    for (c, in_, out_), data in zip(steps, want):
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
    for k, (c, in_, out_) in enumerate(steps[-300:]):
        for o in out_[::11]:
            assert ge.get(o) == oe.get(o), "step %d (%s) wire %d" % (k, c.name, o)
    ctx.sync()
    gg.close(); ge.close(); ctx.close()


@pytest.mark.parametrize("drop_at", ["1", "3", "4"])
def test_stream_cooperative_pass_that_loses_a_workgroup_is_repeated(drop_at, monkeypatch):
    """GC_COOP_FORCE_TIMEOUT=n: in the n-th cooperative pass of the context one workgroup takes no part, the others' bounded
    wait runs out — as when a workgroup does not become resident in time.  The pass is done again on the device before
    anything that follows it: no call fails, every byte and every evaluated label is the oracle's ((*Streaming).Garble never
    fails spuriously, circuit/stream_garble.go:161-192), the context counts the timeout and keeps to level launches from
    then on.  Passes 1 / 3 belong to the garbler (its 1st and 3rd big step), pass 4 to the evaluator's first block."""
    from scripts.bench_stream import make_steps
    monkeypatch.setenv("GC_COOP_FORCE_TIMEOUT", drop_at)
    nin = 256
    steps = make_steps(3, 24, 2048, 0.3, nin)
    prim = list(range(nin))
    for k in range(1, len(steps)):
        prim += [k * nin + i for i in range(steps[k - 1][0].num_outputs, nin)]
    key = drbg("coopdrop", 32)
    rnd = drbg("coopdrop-rnd", 16 * (len(prim) + 1))
    og = oracle.Stream(key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    ctx = engine.Context(0)
    gg, ge, oe = engine.Stream(ctx, key, rnd, prim), engine.StreamEval(ctx, key), oracle.StreamEval(key)
    for w in prim:
        ge.set(w, og.get(w)["l0"])
        oe.set(w, og.get(w)["l0"])
    # all three steps queued before the first bytes are asked for: the steps behind the failed pass are on the stream already
    for c, in_, out_ in steps:
        gg.garble_begin(c.Gates, c.NumWires, in_, out_)
    got = [gg.garble_finish() for _ in steps]
    state, timeouts = ctx.coop_stats()
    if state == 0 or (state < 0 and timeouts == 0):
        pytest.skip("cooperative passes are not in use on this box (self-test / GC_NO_COOP)")
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, "step %d" % k
    for c, in_, out_ in steps:
        for o in out_[::5]:
            assert gg.get(o) == og.get(o)
    for (c, in_, out_), data in zip(steps, want):
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
    for c, in_, out_ in steps:
        for o in out_[::5]:
            assert ge.get(o) == oe.get(o)
    ctx.sync()
    state, timeouts = ctx.coop_stats()
    assert timeouts == 1 and state == -1, (state, timeouts)
    # ... and the context is as good as new: another stream on it (level launches now) matches the oracle
    g2 = engine.Stream(ctx, key, rnd, prim)
    assert [g2.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps] == want
    g2.close(); gg.close(); ge.close(); ctx.close()


@pytest.mark.parametrize("threads,big", [("0", False), ("3", False), ("3", True), ("0", True), ("3", "deep"), ("3", "framed")])
def test_stream_evaluator_survives_hostile_blocks(threads, big, monkeypatch):
    """a slice of tests/hostile_fuzz.py (10 000 mutants logged in profiles/r04_hostile_fuzz.log): mutated OpCircuit blocks — bit
    flips, truncation, id-width / tmp-flag flips, swapped and replaced ids, row bytes only, gate counts off — through the
    skeleton matcher and the gate-by-gate parser; an accepted mutant ends on the oracle's labels, a rejected one is one the
    oracle rejects or the engine refuses by design, and leaves the wire store untouched"""
    from tests import hostile_fuzz
    monkeypatch.setenv("GC_STREAM_THREADS", threads)
    if big == "deep":  # blocks that run on the deep lanes
        n, stats, kinds, (parsed, matched) = hostile_fuzz.run(30, seed=11, deep=True)
    elif big == "framed":  # through gc_stream_eval_blocks, header sizes mutated as well
        n, stats, kinds, (parsed, matched) = hostile_fuzz.run(150, seed=13, framed=True)
        n = stats["accepted"] + stats["rejected"] + stats["rejected_stricter"]  # (mutants that ran on into the tail are not counted)
        big = False
    else:
        n, stats, kinds, (parsed, matched) = hostile_fuzz.run(24 if big else 100, seed=7 + int(threads), big=big)
    assert stats["accepted"] + stats["rejected"] + stats["rejected_stricter"] == n and len(kinds) >= (5 if big else 8)
    assert stats["accepted"] > 0 and stats["rejected"] + stats["rejected_stricter"] > 0
    assert matched > 0 and parsed > 0


@pytest.mark.parametrize("pinned", [False, True])
def test_stream_evaluator_survives_hostile_blocks_in_device_matched_buffers(pinned):
    """tests/hostile_fuzz.py: run_device — the mutants inside read buffers of 64 KiB and more, whose blocks the GPU recognises
    (stream_eval_dev.cpp): guessed boundaries, verdicts and repeat-pattern checks on the device, the host's parser behind them"""
    from tests import hostile_fuzz
    n, stats, kinds, dev = hostile_fuzz.run_device(60, seed=21 + int(pinned), pinned=pinned)
    assert stats["accepted"] > 0 and stats["rejected"] + stats["rejected_stricter"] > 0 and len(kinds) >= 8
    assert dev[0] > 100  # (the buffers really went to the device)


def test_stream_finish_view_hands_out_the_same_bytes(monkeypatch):
    """gc_stream_garble_finish_view (no copy: a pointer into the pinned staging, valid until the next finish) against the oracle
    on a program with grouped, deep and big steps, mixed with copying finishes; the slot a view points into is only given back
    by the next call"""
    monkeypatch.setenv("GC_STREAM_DEEP_STEPS", "100")
    ctx = engine.Context(0)
    steps, prim = _deep_dependency_program(0x20000)
    key = drbg("viewkey", 32)
    rnd = drbg("viewrnd", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    issued = 0
    for k in range(len(steps)):
        while issued < len(steps) and issued < k + 20:
            c, in_, out_ = steps[issued]
            gg.garble_begin(c.Gates, c.NumWires, in_, out_)
            issued += 1
        got = gg.garble_finish() if k % 5 == 4 else gg.garble_finish_view()
        assert got == want[k], "step %d (%s)" % (k, steps[k][0].name)
    for c, in_, out_ in steps[-10:]:
        for o in out_[::9]:
            assert gg.get(o) == og.get(o)
    gg.close(); ctx.close()


def test_stream_deep_lanes_three_contexts_concurrently(monkeypatch):
    """three contexts on three threads run the deep-lane program at once — garbler and evaluator each (lanes, probes and
    buffer lists are per context; what the process shares is the runtime's hardware queues): every stream's bytes and every
    evaluator's labels are the oracle's"""
    import hashlib
    import threading
    monkeypatch.setenv("GC_STREAM_DEEP_STEPS", "100")
    steps, prim = _deep_dependency_program(0x20000)
    key = drbg("c3key", 32)
    rnd = drbg("c3rnd", 16 * (len(prim) + 1))
    og, oe = oracle.Stream(key, rnd, prim), oracle.StreamEval(key)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    labels0 = {w: (int(og.get(w)["l0"]["d0"]), int(og.get(w)["l0"]["d1"])) for w in prim}
    for w in prim:
        oe.set(w, labels0[w])
    for (c, in_, out_), data in zip(steps, want):
        oe.circuit(c.NumGates, c.NumWires, max(max(in_), max(out_)) + 1, data)
    probe = [o for _, _, out_ in steps[-12:] for o in out_[::11]]
    want_eval = [oe.get(o) for o in probe]
    res = [None] * 3

    def run(i):
        try:
            ctx = engine.Context(0)
            gg, ge = engine.Stream(ctx, key, rnd, prim), engine.StreamEval(ctx, key)
            for w in prim:
                ge.set(w, labels0[w])
            issued, ok = 0, True
            for k in range(len(steps)):
                while issued < len(steps) and issued < k + 24:
                    c, in_, out_ = steps[issued]
                    gg.garble_begin(c.Gates, c.NumWires, in_, out_)
                    issued += 1
                b = gg.garble_finish_view() if (k + i) % 2 else gg.garble_finish()
                ok = ok and b == want[k]
                c, in_, out_ = steps[k]
                ok = ok and ge.circuit(c.NumGates, c.NumWires, max(max(in_), max(out_)) + 1, b) == len(b)
            ok = ok and [ge.get(o) for o in probe] == want_eval
            lanes = gg.deep_stats()[1]
            ctx.sync()
            gg.close(); ge.close(); ctx.close()
            res[i] = "ok" if ok else "mismatch (lanes %d)" % lanes
        except Exception as e:  # noqa: BLE001 - reported through the assertion below
            res[i] = "error: %s" % e

    th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert res == ["ok"] * 3, res


def test_host_api_cooperative_pass_that_loses_a_workgroup_is_repeated(monkeypatch):
    """the same stand-by outside the streams: ONE instance of a wide circuit through the host-buffer calls (gc_garble /
    gc_eval: label exchange by separate kernels, no StoreXchg) with the first cooperative pass forced to lose a workgroup —
    R, tables, every wire and the evaluated labels are the oracle's, nothing fails"""
    if os.environ.get("GC_NO_COOP"):
        pytest.skip("the cooperative passes are switched off (GC_NO_COOP)")
    from mpc_amd.circuit import synthetic_levelised
    from tests.test_gpu_garble_eval import check_garble_eval
    monkeypatch.setenv("GC_COOP_FORCE_TIMEOUT", "1")
    ctx = engine.Context(0)
    c = synthetic_levelised(16, 4096, 0.3, seed=77, ninputs=128, inv_frac=0.05, or_frac=0.02)
    check_garble_eval(ctx, c, drbg("coop-host", 32), 1, "coophost", check_all_wires=True, schedule=1)
    state, timeouts = ctx.coop_stats()
    if state != 0:  # (0: the cooperative passes are not in use on this box)
        assert timeouts >= 1 and state == -1, (state, timeouts)
    check_garble_eval(ctx, c, drbg("coop-host", 32), 1, "coophost2", check_all_wires=False, schedule=1)  # level launches now
    ctx.close()


def _framed(steps, blocks):
    """the stream as the peer frames it (compiler/ssa/streamer.go:679-693): OpCircuit, step, numGates, numTmpWires, numWires"""
    import struct
    out, starts = bytearray(), []
    for k, ((c, in_, out_), data) in enumerate(zip(steps, blocks)):
        starts.append(len(out))
        out += struct.pack(">5I", 1, k, c.NumGates, c.NumWires, max(max(in_), max(out_)) + 1) + bytes(data)
    return bytes(out), starts


@pytest.mark.parametrize("threads,piece", [("3", None), ("3", 1 << 20), ("3", 40000), ("0", 70000), ("3", 999)])
def test_stream_eval_blocks_equals_block_by_block(threads, piece, monkeypatch):
    """gc_stream_eval_blocks over the framed stream — in one call, in pieces of a conn read buffer, in pieces shorter than most
    blocks, with and without the thread that compares ahead — leaves the labels of the per-block calls and of the oracle's
    StreamEval on every wire; a piece that ends inside a block reports `more` and nothing of that block is consumed"""
    monkeypatch.setenv("GC_STREAM_DEEP_STEPS", "100")
    monkeypatch.setenv("GC_STREAM_THREADS", threads)
    ctx = engine.Context(0)
    steps, prim = _deep_dependency_program(0x300)
    steps = steps * 3  # (every block form comes back: the later ones are matched by skeleton, ahead of the calling thread)
    key = drbg("blockskey", 32)
    rnd = drbg("blocks", 16 * (len(prim) + 1))
    og = oracle.Stream(key, rnd, prim)
    blocks = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    framed, starts = _framed(steps, blocks)
    ge, gb, oe = engine.StreamEval(ctx, key), engine.StreamEval(ctx, key), oracle.StreamEval(key)
    bits = np.frombuffer(drbg("blocksbits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = og.get(w)
        lab = wire["l1"] if b else wire["l0"]
        for ev in (ge, gb, oe):
            ev.set(w, lab)
    for (c, in_, out_), data in zip(steps, blocks):
        nw = max(max(in_), max(out_)) + 1
        assert ge.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
        assert oe.circuit(c.NumGates, c.NumWires, nw, data) == len(data)
    # the framed stream + an OpReturn word (2) behind it: the call stops in front of it
    stream = framed + b"\x00\x00\x00\x02" + b"\x00\x00\x00\x07"
    pos, done, win, calls = 0, 0, piece or len(stream), 0
    while done < len(steps):
        used, nb, more = gb.blocks(stream[pos:pos + win])
        calls += 1
        assert pos + used in starts + [len(framed)], "stopped inside a block"
        if used == 0:
            assert more, "no progress and no request for more bytes"
            win *= 2
        else:
            win = piece or len(stream)
        pos, done = pos + used, done + nb
        assert calls < 100000
    assert (pos, done) == (len(framed), len(steps))
    assert gb.blocks(stream[pos:]) == (0, 0, False)  # OpReturn: the caller's
    for k, (c, in_, out_) in enumerate(steps):
        for o in out_[::5]:
            want = oe.get(o)
            assert ge.get(o) == want and gb.get(o) == want, "step %d (%s) wire %d" % (k, c.name, o)
    assert sum(gb.stats()) == len(steps) and gb.stats()[1] > 0
    ctx.sync()
    ge.close(); gb.close(); ctx.close()


def test_stream_eval_blocks_refuses_what_the_block_calls_refuse():
    """a bad block in the middle of the buffer: the blocks in front of it are evaluated, the error is the per-block call's,
    *consumed stops at the bad block; a truncated tail is `more`, not an error"""
    ctx = engine.Context(0)
    steps, prim = _dependency_program(0)
    key = drbg("blocksbad", 16)
    rnd = drbg("blocksbadr", 16 * (len(prim) + 1))
    og = oracle.Stream(key, rnd, prim)
    blocks = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    framed, starts = _framed(steps, blocks)

    def fresh():
        ev = engine.StreamEval(ctx, key)
        for w in prim:
            ev.set(w, og.get(w)["l0"])
        return ev

    k = len(steps) // 2
    bad = bytearray(framed)
    bad[starts[k] + 20] = (bad[starts[k] + 20] & 0xf0) | 0x0f  # the first gate's operation: none of XOR .. INV
    ev = fresh()
    ev.blocks(framed)  # (every block form is known now: the second run matches ahead)
    with pytest.raises(engine.EngineError) as ei:
        ev.blocks(bytes(bad))
    assert ei.value.code == engine.GC_E_GATE
    assert ev.last_blocks[:2] == (starts[k], k)
    ev.close()
    ev = fresh()
    cut = starts[k] + 20 + len(blocks[k]) // 2
    assert ev.blocks(framed[:cut]) == (starts[k], k, True)
    assert ev.blocks(framed[starts[k]:]) == (len(framed) - starts[k], len(steps) - k, False)
    good = fresh()
    good.blocks(framed)
    for c, in_, out_ in steps:
        for o in out_[::3]:
            assert ev.get(o) == good.get(o)
    # header sizes that would size arrays by the peer's word: refused before anything is allocated
    import struct
    for ntmp, nwires in ((0xffffffff, 100), (100, 0xffffffff), (64 * 3 + (1 << 20) + 1, 100)):
        ev = fresh()
        with pytest.raises(engine.EngineError) as ei:
            ev.blocks(struct.pack(">5I", 1, 0, 3, ntmp, nwires) + bytes(64))
        assert ei.value.code == engine.GC_E_ARG and ev.last_blocks[:2] == (0, 0)
        with pytest.raises(engine.EngineError) as ei:
            ev.circuit(3, ntmp, nwires, bytes(64))
        assert ei.value.code == engine.GC_E_ARG
        ev.close()
    # a header cut in two, a lone byte
    assert fresh().blocks(framed[:starts[1] + 7]) == (starts[1], 1, True)
    assert fresh().blocks(framed[:1]) == (0, 0, True)
    assert fresh().blocks(b"") == (0, 0, False)
    ctx.sync()
    ctx.close()


def test_stream_released_handles():
    """gc_stream_release: the circuit goes back to the bounded cache, the handle is refused afterwards and its number is handed
    out again; a step queued before the release is unaffected"""
    from mpc_amd.circuit import adder
    ctx = engine.Context(0)
    c1, c2 = adder(16), adder(24)
    prim = list(range(100))
    key, rnd = drbg("relkey", 16), drbg("rel", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    h1 = gg.intern(c1.Gates, c1.NumWires, c1.num_inputs, c1.num_outputs)
    h2 = gg.intern(c2.Gates, c2.NumWires, c2.num_inputs, c2.num_outputs)
    in1, out1 = list(range(c1.num_inputs)), list(range(200, 200 + c1.num_outputs))
    in2, out2 = list(range(40, 40 + c2.num_inputs)), list(range(300, 300 + c2.num_outputs))
    gg.garble_begin_h(h1, in1, out1)
    gg.release(h1)  # (with the step still queued)
    with pytest.raises(engine.EngineError) as ei:
        gg.garble_begin_h(h1, in1, out1)
    assert ei.value.code == engine.GC_E_ARG
    with pytest.raises(engine.EngineError):
        gg.release(h1)
    gg.garble_begin_h(h2, in2, out2)
    assert gg.garble_finish() == og.garble(c1.Gates, c1.NumWires, in1, out1)
    assert gg.garble_finish() == og.garble(c2.Gates, c2.NumWires, in2, out2)
    h3 = gg.intern(c1.Gates, c1.NumWires, c1.num_inputs, c1.num_outputs)  # the same circuit again: found in the cache, a handle of its own
    assert h3 == h1
    gg.garble_begin_h(h3, in1, [o + 500 for o in out1])
    assert gg.garble_finish() == og.garble(c1.Gates, c1.NumWires, in1, [o + 500 for o in out1])
    ctx.sync()
    gg.close(); ctx.close()


def test_c_host_drives_both_sides_of_a_streamed_program():
    """tools/stream_driver.c — a plain C host over include/gcengine.h, the calls a cgo shim makes: garbler by handle with a
    window, evaluator block by block AND the framed stream through gc_stream_eval_blocks in 1 MiB pieces (the driver itself
    compares the two evaluators' labels); the stream's SHA-256 is the oracle's (tests/golden/stream_bench_golden.json)"""
    from scripts import bench_stream
    if not os.path.exists(bench_stream.NATIVE):
        pytest.skip("tools/stream_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = bench_stream.run_native("uniform512", bytes(range(32)), 64)
    assert r["sha256_ok"] is True and r["steps"] == 4000
    assert r["eval_blocks_matched"] > 3900 and r["eval_blocks_gates_per_s"] and r["eval_blocks_chunk"] == 1 << 20


@pytest.mark.parametrize("window", [1, 8])
def test_stream_serialiser_pieces_and_lone_steps(window):
    """the group serialiser writes a job in pieces of 512 gates, each built in LDS at the piece's own alignment: jobs of 40 to
    3 600 gates (one to eight pieces, ragged last pieces, OR / INV / XNOR rows) side by side in one group, global ids on both
    sides of 0xffff (16- and 32-bit id forms inside one job, so the pieces' byte offsets are not multiples of anything);
    window 1: every step alone — kernel, serialiser and copy on one stream (the unchanged caller's path)"""
    from mpc_amd.circuit import synthetic_levelised
    ctx = engine.Context(0)
    shapes = [(2, 20, 0.3, 0.0, 0.0, 0.0), (5, 100, 0.4, 0.1, 0.1, 0.1), (6, 170, 0.5, 0.05, 0.2, 0.0), (12, 300, 0.3, 0.1, 0.1, 0.2),
              (9, 57, 0.9, 0.0, 0.0, 0.0), (4, 128, 0.0, 0.0, 0.5, 0.2)]
    steps, prim, nxt = [], [], 0xfe80
    for i, (lv, w, fa, fo, fi, fx) in enumerate(shapes * 2):
        c = synthetic_levelised(lv, w, fa, seed=900 + i, ninputs=24, or_frac=fo, inv_frac=fi, xnor_frac=fx)
        in_ = list(range(nxt, nxt + 24))
        out_ = list(range(nxt + 24, nxt + 24 + c.num_outputs))
        nxt += 24 + c.num_outputs + 3
        prim += in_
        steps.append((c, in_, out_))
    assert max(c.NumGates for c, _, _ in steps) > 3000 and steps[-1][2][-1] > 0x10000 > steps[0][1][0]
    key = drbg("serkey", 32)
    rnd = drbg("ser", 16 * (len(prim) + 1))
    og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
    want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
    got, issued = [], 0
    for k in range(len(steps)):
        while issued < len(steps) and issued < k + window:
            c, in_, out_ = steps[issued]
            gg.garble_begin(c.Gates, c.NumWires, in_, out_)
            issued += 1
        got.append(gg.garble_finish())
    for k, (w, g) in enumerate(zip(want, got)):
        assert g == w, "step %d of %d (%d gates)" % (k, len(steps), steps[k][0].NumGates)
    groups, grouped, bigs = gg.stats()
    assert grouped == len(steps) and (groups == len(steps) if window == 1 else groups < len(steps))
    for c, in_, out_ in steps:
        for o in out_:
            assert gg.get(o) == og.get(o)
    gg.close(); ctx.close()
