"""Oracle streaming garbler / evaluator (circuit/stream_garble.go, stream_evaluator.go): the serialised
stream of a sequence of per-step circuits evaluates to the plaintext result, wires persist across
circuits through in[] / out[], and the 16-/32-bit index forms are both exercised."""
import numpy as np

import oracle
from mpc_amd.circuit import comparator64, parse_file, synthetic_levelised
from tests.util import drbg


def run_program(garbler, evaluator, steps, rdict):
    """steps: list of (circuit, in wire ids, out wire ids).  Returns the byte streams."""
    streams = []
    for c, in_, out_ in steps:
        data = garbler.garble(c.Gates, c.NumWires, in_, out_)
        streams.append(data)
        if evaluator is not None:
            ntmp = c.NumWires  # the Go driver passes maxID+1 for the tmp store (streamer.go:679-693)
            used = evaluator.circuit(c.NumGates, ntmp, max(max(in_), max(out_)) + 1, data)
            assert used == len(data)
    return streams


def make_program(base=0):
    """three chained steps re-using global wires; `base` shifts global ids (>0xffff -> 32-bit indexes)"""
    c1 = synthetic_levelised(5, 24, 0.3, seed=31, ninputs=16, or_frac=0.1, inv_frac=0.15, xnor_frac=0.1)
    c2 = comparator64()
    c3 = synthetic_levelised(4, 40, 0.5, seed=32, ninputs=c1.num_outputs + 1, inv_frac=0.1)
    g_in1 = [base + i for i in range(16)]
    g_out1 = [base + 100 + i for i in range(c1.num_outputs)]
    g_in2 = [base + 200 + i for i in range(128)]
    g_out2 = [base + 400]
    g_in3 = g_out1 + g_out2  # consumes earlier results
    g_out3 = [base + 500 + i for i in range(c3.num_outputs)]
    prim = g_in1 + g_in2
    return [(c1, g_in1, g_out1), (c2, g_in2, g_out2), (c3, g_in3, g_out3)], prim


def plain_program(steps, prim, bits):
    val = dict(zip(prim, bits))
    for c, in_, out_ in steps:
        w = oracle.compute(c.Gates, c.NumWires, c.num_inputs, np.array([val[i] for i in in_], np.uint8))
        for j, o in enumerate(out_):
            val[o] = int(w[c.NumWires - len(out_) + j])
    return val


def check_program(base, keylen):
    steps, prim = make_program(base)
    key = drbg("skey", keylen)
    rnd = drbg("srnd%d" % base, 16 * (len(prim) + 1))
    g = oracle.Stream(key, rnd, prim)
    e = oracle.StreamEval(key)
    bits = np.frombuffer(drbg("sbits", len(prim)), np.uint8) & 1
    for w, b in zip(prim, bits):
        wire = g.get(w)
        e.set(w, wire["l1"] if b else wire["l0"])
    streams = run_program(g, e, steps, None)
    val = plain_program(steps, prim, bits)
    for c, in_, out_ in steps:
        for o in out_:
            wire = g.get(o)
            got = e.get(o)
            want = wire["l1"] if val[o] else wire["l0"]
            assert got == (int(want["d0"]), int(want["d1"]))
    return streams


def test_stream_roundtrip_short_and_long_indexes():
    s16 = check_program(0, 32)
    s32 = check_program(0x20000, 16)
    # first byte of every gate carries the short flag only in the 16-bit form; sizes differ accordingly
    assert s16[0][0] & 0x10 and not (s32[0][0] & 0x10)
    assert len(s32[0]) > len(s16[0])


def test_stream_tweak_restarts_per_circuit_and_r_first():
    steps, prim = make_program(0)
    key = bytes(range(32))
    rnd = drbg("x", 16 * (len(prim) + 1))
    g = oracle.Stream(key, rnd, prim)
    r = oracle.label_set_s(oracle.label_from_bytes(rnd[:16]), True)
    w0 = g.get(prim[0])
    assert (int(w0["l0"]["d0"]), int(w0["l0"]["d1"])) == oracle.label_from_bytes(rnd[16:32])
    assert (int(w0["l0"]["d0"] ^ w0["l1"]["d0"]), int(w0["l0"]["d1"] ^ w0["l1"]["d1"])) == r
    # garbling the same circuit twice on the same inputs gives the same bytes: id restarts at 0 (:174)
    c, in_, out_ = steps[0]
    assert g.garble(c.Gates, c.NumWires, in_, out_) == g.garble(c.Gates, c.NumWires, in_, out_)


def test_hostile_fuzz_helpers_read_blocks_like_the_reference():
    """tests/hostile_fuzz.py's own reading of an OpCircuit block (the classifier behind the GPU fuzz): it walks the oracle's
    bytes gate by gate, recognises the cases the engine refuses by design, and every mutation operator yields a block"""
    import numpy as np
    from mpc_amd.circuit import adder
    from tests import hostile_fuzz as hf
    from tests.util import drbg
    c = adder(16)
    prim = list(range(40))
    og = oracle.Stream(drbg("hfk", 32), drbg("hfr", 16 * (len(prim) + 1)), prim)
    in_, out_ = list(range(32)), list(range(100, 116))
    data = og.garble(c.Gates, c.NumWires, in_, out_)
    gates, err = hf.parse(data, c.NumGates)
    assert err is None and len(gates) == c.NumGates and gates[-1][6] + 16 * gates[-1][7] == len(data)
    assert sum(q[7] for q in gates) == c.slab_rows()
    nw = 116
    assert hf.stricter(data, c.NumGates, c.NumWires, nw) is None
    assert hf.stricter(data, c.NumGates, c.NumWires, 50) == "global id out of range"
    assert hf.stricter(data, c.NumGates, 3, nw) in ("tmp id out of range", "tmp read before written")
    assert hf.stricter(data[:9], 1000, c.NumWires, nw) == "more gates than bytes"
    assert hf.parse(data[:-1], c.NumGates)[1] == "rows"
    bad = bytearray(data); bad[gates[3][0]] = (bad[gates[3][0]] & 0xf0) | 9
    assert hf.parse(bytes(bad), c.NumGates)[1] == "gate"
    rng = np.random.default_rng(3)
    kinds = set()
    for _ in range(300):
        mut, ng, what = hf.mutate(rng, data, c.NumGates)
        assert isinstance(mut, bytes) and ng >= 1
        kinds.add(what)
    assert len(kinds) >= 10
