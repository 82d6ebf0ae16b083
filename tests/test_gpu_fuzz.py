"""Differential fuzzing of the GPU schedules against the oracle on small random circuits that stress the planner:
long XOR fan-in (term lists over 8 / 32 entries -> lane-split items, second XOR rounds), XOR of a wire with itself,
dead gates, outputs that are input wires or produced by every gate type, re-used wire ids, circuits without any
table-producing gate and without any gate at all."""
import numpy as np
import pytest

import oracle
from mpc_amd import engine
from mpc_amd.circuit import AND, GATE, INV, OR, XNOR, XOR, Circuit
from tests.test_gpu_garble_eval import check_garble_eval

pytestmark = pytest.mark.gpu

KEY = bytes(range(100, 132))


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context(0)
    yield c
    c.close()


def random_circuit(rng, ninputs, ngates, p_xor, reuse, nout):
    """gate i reads two earlier wires (biased to recent ones) and writes either a fresh wire or, with probability
    `reuse`, an existing NON-input wire again (the reference allows re-assignment; the plan renames)."""
    gates = np.zeros(ngates, GATE)
    nw = ninputs
    live = list(range(ninputs))
    for i in range(ngates):
        a = live[int(rng.integers(max(0, len(live) - 12), len(live)))] if rng.random() < 0.7 else live[int(rng.integers(0, len(live)))]
        b = live[int(rng.integers(0, len(live)))]
        u = rng.random()
        if u < p_xor:
            op = XOR if rng.random() < 0.8 else XNOR
            if rng.random() < 0.03:
                b = a  # x ^ x
        else:
            op = [AND, OR, INV][int(rng.integers(0, 3))]
        if nw > ninputs and rng.random() < reuse:
            out = int(rng.integers(ninputs, nw))
        else:
            out = nw
            nw += 1
            live.append(out)
        gates[i] = (a, 0 if op == INV else b, out, op, 0)
    # outputs are the LAST nout wires: append copies (XOR with a zero... not available) -> just make sure enough wires
    nout = min(nout, nw)
    return Circuit(nw, [ninputs // 2, ninputs - ninputs // 2], [nout], gates)


def xor_tree(rng, ninputs, fan):
    """outputs that are XORs of up to `fan` inputs through chains and trees (one chunk, several XOR rounds)"""
    gates = []
    nw = ninputs
    acc = 0
    for i in range(1, fan):  # chain: list length i+1
        gates.append((acc, i % ninputs if i % ninputs != acc else (i + 1) % ninputs, nw, XNOR if i % 7 == 0 else XOR))
        acc = nw
        nw += 1
    # a few ANDs consuming intermediate chain values so that they have to be materialised
    for j in range(0, fan - 1, 5):
        gates.append((ninputs + j, (j * 3) % ninputs, nw, AND))
        nw += 1
    g = np.zeros(len(gates), GATE)
    for k, (a, b, o, op) in enumerate(gates):
        g[k] = (a, b, o, op, 0)
    return Circuit(nw, [ninputs // 2, ninputs - ninputs // 2], [min(8, nw)], g)


@pytest.mark.parametrize("seed", range(12))
def test_random_circuits(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    ninputs = int(rng.integers(2, 40))
    ngates = int(rng.integers(1, 400))
    c = random_circuit(rng, ninputs, ngates, p_xor=float(rng.choice([0.0, 0.5, 0.8, 0.95, 1.0])),
                       reuse=float(rng.choice([0.0, 0.05, 0.2])), nout=int(rng.integers(1, 12)))
    batch = int(rng.choice([1, 3, 17, 70]))
    for schedule in (0, 1, 2):
        check_garble_eval(ctx, c, KEY, batch, "fuzz%d" % seed, schedule=schedule)


@pytest.mark.parametrize("batch", [2100, 4100, 16500])
def test_host_api_large_batches_switch_layout(ctx, batch):
    """gc_garble / gc_eval on pooled batches whose tile (8 - 64 instances per workgroup at these sizes) has to shrink
    when a call keeps every wire (level-walking kernel, more LDS per instance than the flattened one): the arrays are
    re-laid out before anything is written (a 400-circuit fuzz run found gc_eval launching with the other kernel's tile)"""
    rng = np.random.default_rng(9100 + batch)
    c = random_circuit(rng, 22, 2337, p_xor=0.8, reuse=0.02, nout=17)
    sample = sorted(set(list(range(0, batch, 401)) + [batch - 1, batch - 2, 1]))
    for schedule in (1, 2):
        check_garble_eval(ctx, c, KEY, batch, "lb%d" % batch, check_all_wires=False, schedule=schedule, sample=sample)


@pytest.mark.parametrize("fan", [9, 17, 33, 70, 200])
def test_long_xor_lists(ctx, fan):
    c = xor_tree(np.random.default_rng(fan), 24, fan)
    for schedule in (0, 1, 2):
        check_garble_eval(ctx, c, KEY, 9, "xt%d" % fan, schedule=schedule)
    # batches whose tiles hold 2 and 4 instances: the lane-split items join across TI-strided lanes
    for batch in (520, 1030):
        check_garble_eval(ctx, c, KEY, batch, "xtb%d" % fan, schedule=1, sample=[0, 1, 2, 3, 4, 5, batch // 2, batch - 2, batch - 1])


def test_no_gates_and_outputs_are_inputs(ctx):
    g = np.zeros(0, GATE)
    c = Circuit(6, [3, 3], [2], g)  # outputs = input wires 4, 5
    for schedule in (0, 1, 2):
        check_garble_eval(ctx, c, KEY, 5, "nogates", schedule=schedule)


def device_pipeline_case(ctx, seed):
    """one random circuit, random key size, schedule and batch (1 - 16 500 instances) through the device-resident
    pipeline: garble -> select -> (eval | table egress / ingest + eval) -> decode; decoded bits against plaintext
    evaluation, tables of sampled instances against the oracle; raises on a mismatch (tests/ext_fuzz.py runs hundreds)"""
    from tests.test_gpu_garble_eval import oracle_instance, rnd_for
    from tests.util import drbg
    rng = np.random.default_rng(77000 + seed)
    ninputs = int(rng.integers(2, 80))
    ngates = int(rng.integers(1, 3000))
    c = random_circuit(rng, ninputs, ngates, p_xor=float(rng.choice([0.0, 0.3, 0.6, 0.8, 0.9, 0.97, 1.0])),
                       reuse=float(rng.choice([0.0, 0.02, 0.1, 0.3])), nout=int(rng.integers(1, 40)))
    batch = int(rng.choice([1, 3, 64, 130, 520, 1030, 2100, 4100, 16500]))
    key = drbg("xk%d" % seed, int(rng.choice([16, 24, 32])))
    schedule = int(rng.choice([0, 1, 1, 2]))
    dc = engine.DeviceCircuit(ctx, c)
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    for b in (gb, ev):
        b.set_schedule(schedule)
    rnd = rnd_for(c, "xd%d" % seed, batch)
    d_rnd = ctx.to_device(rnd)
    bits = (np.frombuffer(drbg("xb%d" % seed, c.num_inputs * batch), np.uint8) & 1).reshape(batch, c.num_inputs)
    d_bits = ctx.to_device(bits)
    d_out = ctx.zeros((batch, max(c.num_outputs, 1)))
    d_mis = ctx.zeros(1, np.int32)
    gb.garble(key, d_rnd)
    ev.select_inputs(gb, d_bits)
    mode = int(rng.integers(0, 3)) if batch <= 4100 and c.slab_rows() else 0
    if mode == 0:
        ev.eval(key, gb)
    else:  # tables leave the garbler in the driver's wire format (1) / sha2pc's dense form (2) and are ingested again
        nbytes = dc.tables_wire_bytes if mode == 1 else 16 * c.slab_rows()
        stride = (nbytes + 63) // 64 * 64
        d_wire = ctx.zeros(batch * stride)
        d_bad = ctx.zeros(1, np.int32)
        if mode == 1:
            gb.egress_tables(d_wire, stride)
            ev.ingest_tables(d_wire, stride, d_bad)
        else:
            gb.egress_tables_dense(d_wire, stride)
            ev.ingest_tables_dense(d_wire, stride)
        ev.eval(key, ev)
        ctx.sync()
        assert int(d_bad.numpy()[0]) == 0, "ingest flagged a header"
        wire = d_wire.numpy().reshape(batch, stride)
        sl = gb.read_slab()
        for i in (0, batch - 1):
            want = oracle.tables_serialize(c.Gates, sl[i]) if mode == 1 else \
                b"".join(int(x).to_bytes(8, "big") for row in sl[i] for x in (row["d0"], row["d1"]))
            assert wire[i, :nbytes].tobytes() == want, "egress bytes of instance %d (mode %d)" % (i, mode)
    gb.decode(ev, d_out, d_mis)
    ctx.sync()
    assert int(d_mis.numpy()[0]) == 0, "decode mismatches"
    out = d_out.numpy()
    for i in sorted(set(list(range(0, batch, max(1, batch // 7))) + [batch - 1])):
        plain = c.compute_bits(bits[i])
        assert (plain[c.NumWires - c.num_outputs:] == out[i][: c.num_outputs]).all(), "decoded bits of instance %d" % i
    slab, R = gb.read_slab(), gb.read_r()
    for i in sorted(set([0, batch // 2, batch - 1])):
        ref = oracle_instance(c, key, rnd, i)
        assert R[i] == ref["R"] and (slab[i] == ref["slab"]).all(), "tables of instance %d" % i
    gb.close(); ev.close(); dc.close()
    return ninputs, ngates, batch, schedule, len(key)




@pytest.mark.parametrize("seed", range(24))
def test_device_pipeline_random(ctx, seed):
    device_pipeline_case(ctx, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 3, 14, 22, 41, 75, 96, 150])
def test_queued_programs_across_the_scheduling_classes(seed):
    """a slice of tests/ext_fuzz.py's seventh pass (tests/queued_case.py): random programs queued 1 to 300 steps ahead — step
    groups, deep steps and the short steps that follow them on their lanes, big steps; results overwrite live wires, operands
    repeat, in-place updates; 0 to 3 lanes, by handle or content — byte for byte and wire for wire against the oracle, both
    sides.  (Seeds 14, 22 and 75 are the ones an early version of the GENERATOR got wrong: it let later steps read output
    wires that no gate writes, which read (0, R) here and (0, 0) in the reference — INTEGRATION.md.)"""
    from tests.queued_case import run_case
    run_case(seed)
