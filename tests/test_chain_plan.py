"""Chain fusion, host side (mpc_amd/csrc/stream_fuse.cpp through gc_plan_create_chain; no GPU): the merged plan of a chain
of streamed circuits keeps what the reference derives from its serial order — the hash tweak starting over at every circuit
(circuit/stream_garble.go:174), the table rows counting through in gate order — and computes, on plaintext bits walked
through the exact unit program the kernels execute, what the steps compute one after the other."""
import numpy as np
import pytest

from mpc_amd import engine
from mpc_amd.circuit import adder, bitwise, multiplier, subtractor, synthetic_levelised

STORE = 0xFFFFFFFF
TWEAK = {0: 0, 1: 0, 2: 2, 3: 1, 4: 1}
ROWS = {0: 0, 1: 0, 2: 2, 3: 3, 4: 1}


def run_steps(steps, store_bits):
    """the reference's order: step after step, every input from the store or from an earlier step's outputs"""
    outs, ext = [], iter(store_bits)
    for c, wiring in steps:
        bits = []
        for i in range(c.num_inputs):
            src = STORE if wiring is None else int(wiring[i])
            bits.append(next(ext) if src == STORE else outs[src >> 24][src & 0xFFFFFF])
        outs.append([int(b) for b in c.compute_bits(bits)[c.NumWires - c.num_outputs:]])
    return [b for o in outs for b in o]


def chain_plan(steps):
    return engine.Plan.chain([(c.Gates, c.NumWires, c.num_inputs, c.num_outputs, w) for c, w in steps])


def n_store_inputs(steps):
    return sum(c.num_inputs if w is None else int((np.asarray(w) == STORE).sum()) for c, w in steps)


def check(steps, seed):
    p = chain_plan(steps)
    # tweaks restart per step, rows count through (garble.go:357-359,419-420,451-452; :199-211; stream_garble.go:174)
    row = 0
    for k, (c, _) in enumerate(steps):
        idv = 0
        for i, g in enumerate(c.Gates):
            m = int(p.gate_base[k]) + i
            assert p.tweak_of_gate[m] == idv and p.row_of_gate[m] == row, (k, i)
            idv += TWEAK[int(g["op"])]
            row += ROWS[int(g["op"])]
    assert p.row_of_gate[-1] == row == sum(c.slab_rows() for c, _ in steps)
    assert p.info.ninputs == n_store_inputs(steps)
    assert p.info.noutputs == sum(c.num_outputs for c, _ in steps)
    rng = np.random.default_rng(seed)
    for _ in range(4):
        bits = rng.integers(0, 2, p.info.ninputs).astype(np.uint8)
        assert p.simulate(bits).tolist() == run_steps(steps, bits.tolist())
    return p


def from_step(m, nbits, first=0):
    return [(m << 24) | (first + j) for j in range(nbits)]


def test_add_chain_is_pipelined_across_the_steps():
    """h = p0 + p1 + ... + p9 as nine 64-bit adders (the sum of ten products of an Ed25519 field multiplication,
    ed25519.mpcl:388-439): bit i of link k + 1 needs bit i of link k, so the fused chain is 63 + 8 hash phases deep, not 9 x 63"""
    add = adder(64)
    steps = [(add, None)] + [(add, from_step(k, 64) + [STORE] * 64) for k in range(8)]
    p = check(steps, 1)
    one = engine.Plan(add.Gates, add.NumWires, add.num_inputs, add.num_outputs)
    assert one.info.n_hash_phases >= 63
    assert p.info.n_flat_steps < 2 * one.info.n_flat_steps  # (nine links in less than two)
    assert p.info.n_flat_slots != 0xFFFFFFFF


def test_values_on_the_plaintext():
    """the fused add chain really adds: sum of four 16-bit values modulo 2^16"""
    add = adder(16)
    steps = [(add, None), (add, from_step(0, 16) + [STORE] * 16), (add, [STORE] * 16 + from_step(1, 16))]
    p = chain_plan(steps)
    vals = [40000, 30000, 1234, 65535]
    bits = [(v >> i) & 1 for v in vals for i in range(16)]
    out = p.simulate(np.array(bits, np.uint8)).tolist()
    got = [sum(b << i for i, b in enumerate(out[16 * k:16 * k + 16])) for k in range(3)]
    assert got == [(40000 + 30000) & 0xFFFF, (40000 + 30000 + 1234) & 0xFFFF, (40000 + 30000 + 1234 + 65535) & 0xFFFF]


def test_mixed_chain_with_partial_wiring_and_every_gate_type():
    """a multiplier at the head, then steps that take some inputs from earlier steps (not only the one before), some from
    the store, out of order and repeated; OR / INV / XNOR gates in the tail circuits"""
    mul, add, sub = multiplier(16), adder(16), subtractor(16)
    syn = synthetic_levelised(4, 16, 0.3, seed=3, ninputs=32, or_frac=0.15, inv_frac=0.15, xnor_frac=0.1)
    x = bitwise(16, 0)
    w1 = from_step(0, 16) + [STORE] * 16                       # add(mul, store)
    w2 = [STORE] * 8 + from_step(1, 8, 4) + from_step(0, 16)   # sub(store | bits 4..11 of the sum, the product again)
    w3 = from_step(2, 16)[::-1] + from_step(1, 16)             # synthetic(reversed difference, sum)
    w4 = [(3 << 24) | (j % syn.num_outputs) for j in range(16)] + [STORE] * 16
    steps = [(mul, None), (add, w1), (sub, w2), (syn, w3), (x, w4)]
    check(steps, 2)


def test_tail_reading_only_the_store_and_a_step_read_twice():
    add = adder(8)
    steps = [(add, None), (add, None), (add, from_step(0, 8) + from_step(0, 8)), (add, from_step(1, 8) + from_step(2, 8))]
    check(steps, 3)


def test_bad_wiring_is_refused():
    add = adder(8)
    with pytest.raises(engine.EngineError):  # names a later step
        chain_plan([(add, None), (add, from_step(1, 8) + [STORE] * 8)])
    with pytest.raises(engine.EngineError):  # names an output the step does not have
        chain_plan([(add, None), (add, from_step(0, 8, first=add.num_outputs) + [STORE] * 8)])
