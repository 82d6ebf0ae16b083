"""Oracle garble -> eval -> decode on the reference's circuits, pinned by the reference's
own golden values that do not depend on Go's random streams:
  sha2pc/sha2pc_test.go:124   expFinal digest (a[i]=i, b[i]=32-i)
  sha2pc/params.go:26         garbledTableLabelCount = 42914
  sha2pc/sha2pc_test.go:239   round3Len = 707146
and by FIPS-197 C.1 for aes_128 (wire order of pkg/crypto/aes/circuit.mpcl:80).
"""
import hashlib

import numpy as np
import pytest

import oracle
from mpc_amd.circuit import AND, INV, OR, XNOR, XOR, GATE, Circuit, and_chain, comparator64, synthetic_levelised
from tests.util import bits_lsb, bits_to_bytes_little, bytes_to_bits_little, drbg, int_from_bits


def garble_eval_decode(c, key, seed, in_bits):
    """One instance through the oracle: returns (decoded output bits, garbled dict, eval wires)."""
    nin = c.num_inputs
    rnd = drbg(seed, 16 * (nin + 1))
    g = oracle.garble(c.Gates, c.NumWires, nin, key, rnd)
    w = np.zeros(c.NumWires, oracle.LABEL)
    for i in range(nin):
        w[i] = g["wires"][i]["l1"] if in_bits[i] else g["wires"][i]["l0"]
    oracle.eval_(c.Gates, c.NumWires, key, w, g["slab"])
    out = []
    for i in range(c.NumWires - c.num_outputs, c.NumWires):
        if w[i] == g["wires"][i]["l0"]:
            out.append(0)
        elif w[i] == g["wires"][i]["l1"]:
            out.append(1)
        else:
            raise AssertionError("unknown label on wire %d" % i)  # circuit.BitFromLabel
    return np.array(out, np.uint8), g, w


def test_garble_rand_order_and_free_xor_invariant(add64_circ):
    c = add64_circ
    key = bytes(range(32))
    rnd = drbg("r0", 16 * (c.num_inputs + 1))
    g = oracle.garble(c.Gates, c.NumWires, c.num_inputs, key, rnd)
    # R = first 16 bytes, S bit forced (garble.go:253-258)
    r = oracle.label_set_s(oracle.label_from_bytes(rnd[:16]), True)
    assert (int(g["R"]["d0"]), int(g["R"]["d1"])) == r
    for i in range(c.num_inputs):  # then one L0 per input wire (garble.go:271-278)
        l0 = oracle.label_from_bytes(rnd[16 * (i + 1) : 16 * (i + 2)])
        assert (int(g["wires"][i]["l0"]["d0"]), int(g["wires"][i]["l0"]["d1"])) == l0
    w = g["wires"]
    assert ((w["l0"]["d0"] ^ w["l1"]["d0"]) == g["R"]["d0"]).all()
    assert ((w["l0"]["d1"] ^ w["l1"]["d1"]) == g["R"]["d1"]).all()
    assert len(g["slab"]) == c.slab_rows()
    # short random stream -> error like a failing io.Reader
    with pytest.raises(oracle.OracleError) as e:
        oracle.garble(c.Gates, c.NumWires, c.num_inputs, key, rnd[:-1])
    assert e.value.code == oracle.E_RAND


@pytest.mark.parametrize("keylen", [16, 24, 32])
def test_all_gate_types_truth_tables(keylen):
    # one gate of each type over 2 input bits; every input combination; many label draws so
    # that all permute-bit cases of OR/INV/AND are hit
    key = drbg("gk", keylen)
    for op in (XOR, XNOR, AND, OR, INV):
        gates = np.zeros(1, GATE)
        gates[0] = (0, 0 if op == INV else 1, 2, op, 0)
        c = Circuit(3, [1, 1], [1], gates)
        seen_perm = set()
        for s in range(24):
            for a in (0, 1):
                for b in (0, 1):
                    out, g, _ = garble_eval_decode(c, key, "tt%d_%d" % (op, s), [a, b])
                    want = {XOR: a ^ b, XNOR: 1 - (a ^ b), AND: a & b, OR: a | b, INV: 1 - a}[op]
                    assert out[0] == want
            seen_perm.add((int(g["wires"][0]["l0"]["d0"]) >> 63, int(g["wires"][1]["l0"]["d0"]) >> 63))
            rows = {XOR: 0, XNOR: 0, AND: 2, OR: 3, INV: 1}[op]
            assert len(g["slab"]) == rows
        assert len(seen_perm) == 4


def test_invalid_gate_type():
    gates = np.zeros(1, GATE)
    gates[0] = (0, 1, 2, 7, 0)
    with pytest.raises(oracle.OracleError) as e:
        oracle.garble(gates, 3, 2, bytes(16), drbg("x", 48))
    assert e.value.code == oracle.E_GATE


def test_eval_corrupted_rows():
    c = and_chain(4)
    g = oracle.garble(c.Gates, c.NumWires, 2, bytes(16), drbg("x", 48))
    w = np.zeros(c.NumWires, oracle.LABEL)
    w[:2] = g["wires"][:2]["l0"]
    with pytest.raises(oracle.OracleError) as e:  # eval.go:54-56
        oracle.eval_(c.Gates, c.NumWires, bytes(16), w, g["slab"][:-1])
    assert e.value.code == oracle.E_ROWS


def test_comparator64_millionaire():
    # README.md:57-108: millionaire example, a=750000 / 900000 vs b=800000
    c = comparator64()
    key = bytes(range(32))
    for a, b in ((750000, 800000), (900000, 800000), (800000, 800000), (0, 0), (2**64 - 1, 2**64 - 2)):
        bits = np.concatenate([bits_lsb(a, 64), bits_lsb(b, 64)])
        plain = oracle.compute(c.Gates, c.NumWires, 128, bits)
        assert plain[-1] == (1 if a > b else 0)
        out, _, _ = garble_eval_decode(c, key, "mill", bits)
        assert out[0] == (1 if a > b else 0)


def test_add64(add64_circ):
    c = add64_circ
    for a, b in ((1, 2), (2**63, 2**63), (0xDEADBEEF12345678, 0x0123456789ABCDEF)):
        bits = np.concatenate([bits_lsb(a, 64), bits_lsb(b, 64)])
        out, _, _ = garble_eval_decode(c, bytes(16), "add", bits)
        assert int_from_bits(out) == (a + b) % (1 << c.num_outputs)


def test_aes128_fips197(aes_circ):
    c = aes_circ
    assert (c.NumGates, c.NumWires, c.num_inputs, c.num_outputs) == (36663, 36919, 256, 128)
    s = c.stats()
    assert (s["AND"], s["INV"], s["XOR"], s["OR"], s["XNOR"]) == (6400, 2087, 28176, 0, 0)
    assert c.slab_rows() == 14887
    key = int.from_bytes(bytes(range(16)), "big")
    pt = int.from_bytes(bytes.fromhex("00112233445566778899aabbccddeeff"), "big")
    bits = np.concatenate([bits_lsb(key, 128), bits_lsb(pt, 128)])
    plain = oracle.compute(c.Gates, c.NumWires, 256, bits)
    assert int_from_bits(plain[-128:]).to_bytes(16, "big").hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"
    for gkey in (bytes(range(32)), b"0123456789abcdef"):  # Garbler's AES-256 key; bench AES-128 key
        out, g, _ = garble_eval_decode(c, gkey, "aes", bits)
        assert int_from_bits(out).to_bytes(16, "big").hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"


def test_aes128_levels(aes_circ):
    # AssignLevels(TargetYao) (circuit.go:206-254); SURVEY §8: 308 levels, max width 192
    g, nl, mw = oracle.assign_levels(aes_circ.Gates, aes_circ.NumWires)
    assert (nl, mw) == (308, 192)


def test_sha256xor_reference_digest(sha_circ):
    c = sha_circ
    assert (c.NumGates, c.NumWires, c.num_inputs, c.num_outputs) == (127806, 128318, 512, 256)
    assert c.slab_rows() == 42914  # sha2pc/params.go:26 garbledTableLabelCount
    # Round-3 payload: "R3"(2) + sid(32) + key(32) + tables + 256 garbler labels + 256 output wires(32 B)
    # + 256 CO ciphertext pairs — sha2pc_test.go:239 round3Len; only the table part depends on this path
    a = bytes(range(32))
    b = bytes(32 - i for i in range(32))
    bits = np.concatenate([bytes_to_bits_little(a), bytes_to_bits_little(b)])
    want = hashlib.sha256(bytes(x ^ y for x, y in zip(a, b))).hexdigest()
    assert want == "4b2f74579fc7c778745121996f604371a326dc5174f9851706032626668abf2e"  # sha2pc_test.go:124
    out, g, _ = garble_eval_decode(c, drbg("shakey", 32), "sha", bits)
    assert bits_to_bytes_little(out).hex() == want
    assert len(g["slab"]) == 42914


def test_synthetic_matches_plaintext():
    c = synthetic_levelised(12, 40, 0.3, seed=3, ninputs=32, or_frac=0.1, inv_frac=0.1, xnor_frac=0.1)
    s = c.stats()
    assert all(s[k] > 0 for k in s)
    for t in range(4):
        bits = np.frombuffer(drbg("sb%d" % t, 32), np.uint8) & 1
        plain = oracle.compute(c.Gates, c.NumWires, 32, bits)
        out, _, _ = garble_eval_decode(c, drbg("sk", 24), "s%d" % t, bits)
        assert (out == plain[-c.num_outputs:]).all()


def test_host_compute_bits_matches_oracle(aes_circ, add64_circ):
    """mpc_amd.circuit.Circuit.compute_bits (host mirror of circuit/computer.go) against the oracle's restatement"""
    import numpy as np
    from mpc_amd.circuit import comparator64, synthetic_levelised
    rng = np.random.default_rng(3)
    for c in (aes_circ, add64_circ, comparator64(),
              synthetic_levelised(6, 40, 0.3, seed=5, ninputs=32, or_frac=0.1, inv_frac=0.1, xnor_frac=0.1)):
        for _ in range(3):
            b = rng.integers(0, 2, c.num_inputs).astype(np.uint8)
            assert (c.compute_bits(b) == oracle.compute(c.Gates, c.NumWires, c.num_inputs, b)).all()
