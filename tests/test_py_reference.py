"""oracle/*.c against a SECOND restatement of the same Go text (tests/py_reference.py: pure Python, written separately, with
its own AES from FIPS-197), on what no vector held by the reference can pin: XNOR, OR- and INV-dense circuits, 16- and 24-byte
keys, the streaming wire format with 16- and 32-bit ids, tmp wires and outputs overwritten in place.  Two restatements that
were written apart and agree on every byte are the next best thing to the Go toolchain the image lacks (DESIGN.md §6 lists
which rows are pinned by Go's own output and which by the two of them)."""
import numpy as np
import pytest

import oracle
from mpc_amd.circuit import adder, comparator64, synthetic_levelised
from tests import py_reference as py
from tests.test_oracle_stream import make_program
from tests.util import drbg


def test_python_aes_is_fips_197():
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    for key, ct in ((bytes(range(16)), "69c4e0d86a7b0430d8cdb78070b4c55a"),           # C.1
                    (bytes(range(24)), "dda97ca4864cdfe06eaf70a0ec0d7191"),           # C.2
                    (bytes(range(32)), "8ea2b7ca516745bfeafc49904b496089")):          # C.3
        assert py.AES(key).encrypt(pt).hex() == ct
    assert py.AES(bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")).encrypt(
        bytes.fromhex("3243f6a8885a308d313198a2e0370734")).hex() == "3925841d02dc09fbdc118597196a0b32"  # App. B


def _gl(c):
    return [(int(g["in0"]), int(g["in1"]), int(g["out"]), int(g["op"])) for g in c.Gates]


def _lab(rec):
    return (int(rec["d0"]), int(rec["d1"]))


CIRCS = [synthetic_levelised(8, 24, 0.3, seed=5, ninputs=20, or_frac=0.15, inv_frac=0.15, xnor_frac=0.2),   # every gate type
         synthetic_levelised(5, 30, 0.0, seed=6, ninputs=16, or_frac=0.5, inv_frac=0.3, xnor_frac=0.2),    # OR / INV dense, no AND
         adder(16)]


@pytest.mark.parametrize("keylen", [16, 24, 32])
@pytest.mark.parametrize("ci", range(len(CIRCS)))
def test_garble_and_eval_agree(ci, keylen):
    c = CIRCS[ci]
    key = drbg("py-key%d" % keylen, keylen)
    for inst in range(2):
        rnd = drbg("py-rnd/%d/%d" % (ci, inst), 16 * (c.num_inputs + 1))
        o = oracle.garble(c.Gates, c.NumWires, c.num_inputs, key, rnd)
        r, wires, tables = py.garble(_gl(c), c.NumWires, c.num_inputs, key, rnd)
        assert _lab(o["R"]) == r
        rows = [row for t in tables for row in t]
        assert len(rows) == len(o["slab"]) == c.slab_rows()
        assert [_lab(x) for x in o["slab"]] == rows                    # every table label, in gate order
        for w in range(c.NumWires):                                     # every wire, both labels
            assert (_lab(o["wires"]["l0"][w]), _lab(o["wires"]["l1"][w])) == wires[w], w
        # evaluation on chosen inputs: the label of the plaintext value on every wire
        bits = np.frombuffer(drbg("py-bits/%d/%d" % (ci, inst), c.num_inputs), np.uint8) & 1
        lab = [None] * c.NumWires
        ow = np.zeros(c.NumWires, oracle.LABEL)
        for i, b in enumerate(bits):
            lab[i] = wires[i][int(b)]
            ow[i] = o["wires"]["l1" if b else "l0"][i]
        py.evaluate(_gl(c), key, lab, tables)
        oracle.eval_(c.Gates, c.NumWires, key, ow, o["slab"])
        plain = c.compute_bits(bits)
        for w in range(c.NumWires):
            assert _lab(ow[w]) == lab[w] == wires[w][int(plain[w])], w


@pytest.mark.parametrize("base,keylen", [(0, 32), (0x20000, 16), (0xfff0, 24)])
def test_streaming_wire_format_agrees(base, keylen):
    """16-bit ids, 32-bit ids, and ids straddling 0xffff (per gate: short form only if all of the gate's ids fit)"""
    steps, prim = make_program(base)
    extra = synthetic_levelised(4, 20, 0.2, seed=9, ninputs=12, or_frac=0.2, inv_frac=0.2, xnor_frac=0.2)
    steps = list(steps) + [(extra, [prim[(3 * i) % len(prim)] for i in range(12)], [base + 0x300 + i for i in range(extra.num_outputs)]),
                           # an in-place update: the outputs overwrite wires the step reads
                           (adder(8), prim[:16], prim[:8])]
    key = drbg("py-skey", keylen)
    rnd = drbg("py-srnd%d" % base, 16 * (len(prim) + 1))
    og, pg = oracle.Stream(key, rnd, prim), py.Stream(key, rnd, prim)
    oe, pe = oracle.StreamEval(key), py.StreamEval(key)
    for w in prim:
        l0 = _lab(og.get(w)["l0"])
        assert l0 == pg.wire(w)[0]
        oe.set(w, l0)
        pe.set(w, l0)
    for c, in_, out_ in steps:
        want = og.garble(c.Gates, c.NumWires, in_, out_)
        got = pg.garble(_gl(c), c.NumWires, list(in_), list(out_))
        assert got == bytes(want)                                       # the bytes on the wire
        for o in out_:
            assert (_lab(og.get(o)["l0"]), _lab(og.get(o)["l1"])) == pg.wire(o)
        nw = max(max(in_), max(out_)) + 1
        assert oe.circuit(c.NumGates, c.NumWires, nw, want) == pe.circuit(c.NumGates, got) == len(got)
        for o in out_:
            assert oe.get(o) == pe.get(o) and pe.get(o) in pg.wire(o)   # one of the garbler's two labels of the wire


def test_comparator_decodes():
    """config 1's stand-in through the Python restatement alone: garble, evaluate, decode a > b"""
    c = comparator64()
    key = drbg("py-cmp", 32)
    rnd = drbg("py-cmp-r", 16 * (c.num_inputs + 1))
    r, wires, tables = py.garble(_gl(c), c.NumWires, c.num_inputs, key, rnd)
    for a, b in ((750000, 800000), (900000, 800000)):
        bits = [(a >> i) & 1 for i in range(64)] + [(b >> i) & 1 for i in range(64)]
        lab = [None] * c.NumWires
        for i, v in enumerate(bits):
            lab[i] = wires[i][v]
        py.evaluate(_gl(c), key, lab, tables)
        out = lab[c.NumWires - 1]
        assert out in wires[c.NumWires - 1] and wires[c.NumWires - 1].index(out) == int(a > b)
