"""GPU streaming garbler (gc_stream_*, config 5 shape): the byte stream must equal the oracle's
restatement of circuit/stream_garble.go byte for byte, for a sequence of per-step circuits."""
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
    (so does every wire afterwards); a third begin is refused; the evaluator, whose calls return before the GPU is done,
    decodes the same stream to the oracle's labels"""
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
            if k + 2 < len(prog):  # two in flight: a third is refused and changes nothing
                with pytest.raises(engine.EngineError) as e:
                    gg.garble_begin(prog[k + 2][0].Gates, prog[k + 2][0].NumWires, prog[k + 2][1], prog[k + 2][2])
                assert e.value.code == engine.GC_E_ARG
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
    import os
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
