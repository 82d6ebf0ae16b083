"""GPU table egress / ingest in the driver's wire format (garbler.go:69-82, evaluator.go:40-66) against
the oracle's restatement, both schedules; ingest(egress(x)) evaluates correctly; corrupted headers are
counted."""
import numpy as np
import pytest

import oracle
from mpc_amd import engine
from mpc_amd.circuit import synthetic_levelised
from tests.util import drbg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("schedule", [0, 1])
def test_egress_ingest(schedule, aes_circ):
    ctx = engine.Context(0)
    for c, batch in ((synthetic_levelised(8, 50, 0.3, seed=41, ninputs=32, or_frac=0.1, inv_frac=0.1, xnor_frac=0.05), 37),
                     (aes_circ, 5)):
        dc = engine.DeviceCircuit(ctx, c)
        gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
        gb.set_schedule(schedule); ev.set_schedule(schedule)
        key = bytes(range(32))
        stride_rnd = 16 * (c.num_inputs + 1)
        rnd = drbg("eg%d" % batch, stride_rnd * batch)
        d_rnd = ctx.to_device(rnd)  # gc_dev_alloc + gc_dev_upload: the buffers come from the C ABI, not from torch
        gb.garble(key, d_rnd)
        nbytes = dc.tables_wire_bytes
        assert nbytes == 4 + 4 * c.NumGates + 16 * c.slab_rows()
        stride = (nbytes + 63) // 64 * 64
        d_wire = ctx.zeros(batch * stride)
        gb.egress_tables(d_wire, stride)
        wire = d_wire.numpy().reshape(batch, stride)
        slab = gb.read_slab()
        for i in range(batch):
            want = oracle.tables_serialize(c.Gates, slab[i])
            assert wire[i, :nbytes].tobytes() == want
            if i < 3:  # and the slab itself is the oracle's
                ref = oracle.garble(c.Gates, c.NumWires, c.num_inputs, key, rnd[i * stride_rnd:(i + 1) * stride_rnd])
                assert (slab[i] == ref["slab"]).all()
                assert (oracle.tables_parse(c.Gates, want) == ref["slab"]).all()
        # evaluator side: ingest the bytes, evaluate, decode
        d_bad = ctx.zeros(1, np.int32)
        ev.ingest_tables(d_wire, stride, d_bad)
        bits = (np.frombuffer(drbg("egb", batch * c.num_inputs), np.uint8) & 1).reshape(batch, -1)
        d_bits = ctx.to_device(bits)
        d_out = ctx.zeros((batch, c.num_outputs))
        d_mis = ctx.zeros(1, np.int32)
        ev.select_inputs(gb, d_bits)
        ev.eval(key, ev)  # evaluator's OWN ingested tables
        gb.decode(ev, d_out, d_mis)
        assert int(d_bad.numpy()[0]) == 0 and int(d_mis.numpy()[0]) == 0
        out = d_out.numpy()
        for i in range(batch):
            plain = oracle.compute(c.Gates, c.NumWires, c.num_inputs, bits[i])
            assert (out[i] == plain[c.NumWires - c.num_outputs:]).all()
        # corrupted headers are counted: wrong gate count in instance 0, wrong row count of gate 0 in instance 1
        w2 = wire.copy()
        w2[0, 3] ^= 1
        w2[min(1, batch - 1), 7] ^= 1
        d_w2 = ctx.to_device(w2.reshape(-1))
        d_bad.zero()
        ev.ingest_tables(d_w2, stride, d_bad)
        assert int(d_bad.numpy()[0]) == 2
        gb.close(); ev.close(); dc.close()
    ctx.close()


def test_dense_encoding_sha2pc(sha_circ):
    """sha2pc/encoding.go:363-411: rows in gate order, BE(D0)||BE(D1) each, garbledTableByteLen = 16 * 42914"""
    c = sha_circ
    batch = 3
    ctx = engine.Context(0)
    dc = engine.DeviceCircuit(ctx, c)
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    key = bytes(range(32))
    stride_rnd = 16 * (c.num_inputs + 1)
    rnd = drbg("dense", stride_rnd * batch)
    d_rnd = ctx.to_device(rnd)
    nbytes = 16 * c.slab_rows()
    assert nbytes == 16 * 42914
    d_wire = ctx.zeros(batch * nbytes)
    gb.garble(key, d_rnd)
    gb.egress_tables_dense(d_wire, nbytes)
    wire = d_wire.numpy().reshape(batch, nbytes)
    ref = oracle.garble(c.Gates, c.NumWires, c.num_inputs, key, rnd[:stride_rnd])
    want = b"".join(oracle.label_to_bytes(l) for l in ref["slab"])
    assert wire[0].tobytes() == want
    slab = gb.read_slab()
    for i in range(batch):
        assert wire[i].tobytes() == b"".join(oracle.label_to_bytes(l) for l in slab[i])
    ev.ingest_tables_dense(d_wire, nbytes)
    ctx.sync()
    assert (ev.read_slab() == slab).all()
    gb.close(); ev.close(); dc.close(); ctx.close()


def test_host_wire_calls_match_oracle(aes_circ):
    """gc_garble_wire / gc_eval_wire (host buffers): the bytes circuit.Garbler's send loop would put on the connection
    (garbler.go:69-82) == the oracle's serialiser over the oracle's slab; ingest + eval of them == oracle Eval;
    a corrupted header is refused (evaluator.go:44-47, eval.go:54-56)"""
    ctx = engine.Context(0)
    key = bytes(range(32))
    for c, batch in ((synthetic_levelised(8, 50, 0.3, seed=43, ninputs=32, or_frac=0.1, inv_frac=0.1, xnor_frac=0.05), 21),
                     (aes_circ, 3)):
        dc = engine.DeviceCircuit(ctx, c)
        nin, nout = c.num_inputs, c.num_outputs
        srnd = 16 * (nin + 1)
        rnd = drbg("hw%d" % batch, srnd * batch)
        g = dc.garble_wire(key, rnd, batch=batch)
        nbytes = dc.tables_wire_bytes
        bits = (np.frombuffer(drbg("hwb", batch * nin), np.uint8) & 1).reshape(batch, -1).astype(bool)
        inputs = np.where(bits, g["io"][:, :nin]["l1"], g["io"][:, :nin]["l0"])
        out = dc.eval_wire(key, g["wire"], inputs, batch=batch)
        for i in range(batch):
            ref = oracle.garble(c.Gates, c.NumWires, nin, key, rnd[i * srnd:(i + 1) * srnd])
            assert g["R"][i] == ref["R"]
            assert g["wire"][i, :nbytes].tobytes() == oracle.tables_serialize(c.Gates, ref["slab"])
            assert (g["io"][i][:nin] == ref["wires"][:nin]).all() and (g["io"][i][nin:] == ref["wires"][c.NumWires - nout:]).all()
            w = np.zeros(c.NumWires, engine.LABEL)
            w[:nin] = inputs[i]
            oracle.eval_(c.Gates, c.NumWires, key, w, ref["slab"])
            assert (out[i] == w[c.NumWires - nout:]).all()
        bad = g["wire"].copy()
        bad[batch - 1, 3] ^= 1  # gate count of the last instance
        with pytest.raises(engine.EngineError) as e:
            dc.eval_wire(key, bad, inputs, batch=batch)
        assert e.value.code == engine.GC_E_ROWS
        dc.close()
    ctx.close()
