"""BASELINE config 3 end to end on the GPU: sha2pc's sha256xor circuit x 256 instances, the evaluator's 256 input
labels per instance delivered by IKNP + COT (65 536 OTs = 128 full chunks), evaluated and decoded.
Checks: IKNP correlation, delivered label == the chosen wire label, decoded digest of every
instance == SHA-256(a xor b) (instance 0 is the reference's own vector, sha2pc/sha2pc_test.go:124), and the
garbled-table label count of sha2pc/params.go:26."""
import hashlib

import numpy as np
import pytest

import oracle
from mpc_amd import engine
from mpc_amd.circuit import LABEL, WIRE
from tests.test_gpu_ot import base_setup
from tests.util import drbg

pytestmark = pytest.mark.gpu

KEY = bytes(range(32))


def bits_of_bytes(bs):  # wire 8*idx+bit = bit `bit` (LSB first) of byte idx (sha2pc/bits.go:4-15)
    return np.unpackbits(np.frombuffer(bytes(bs), np.uint8), bitorder="little")


def test_sha256xor_batch_with_iknp_cot(sha_circ):
    c = sha_circ
    batch = 256
    assert c.slab_rows() == 42914  # garbledTableLabelCount, sha2pc/params.go:26
    ctx = engine.Context(0)
    dc = engine.DeviceCircuit(ctx, c)
    a = [bytes(range(32))] + [drbg("a%d" % i, 32) for i in range(1, batch)]
    b = [bytes(32 - i for i in range(32))] + [drbg("b%d" % i, 32) for i in range(1, batch)]
    rnd = drbg("cfg3", 16 * (c.num_inputs + 1) * batch)
    g = dc.garble(KEY, rnd, batch=batch, want_wires=False, want_io=True)
    nin_g, nin_e = c.Inputs[0], c.Inputs[1]
    io = g["io"]  # [batch][inputs + outputs] wires
    # garbler's own input labels (sent in the clear, garbler.go:85-100)
    a_bits = np.stack([bits_of_bytes(x) for x in a]).astype(bool)
    b_bits = np.stack([bits_of_bytes(x) for x in b]).astype(np.uint8)
    inputs = np.zeros((batch, c.num_inputs), LABEL)
    inputs[:, :nin_g] = np.where(a_bits, io[:, :nin_g]["l1"], io[:, :nin_g]["l0"])
    # evaluator's labels by OT: IKNP extension + COT (garbler.go:102-132, cot.go:136-235)
    n = batch * nin_e
    assert n == 65536
    flags = b_bits.reshape(-1)
    wires = np.ascontiguousarray(io[:, nin_g:nin_g + nin_e].reshape(-1), dtype=WIRE)
    base, delta, k0 = base_setup("cfg3")
    rcv, snd = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    u, got = rcv.receive(flags)
    assert len(u) == 128 * 8192  # 128 full chunks
    data = snd.send(u, n)
    x0 = data["d0"] ^ np.where(flags == 1, np.uint64(delta[0]), np.uint64(0))
    x1 = data["d1"] ^ np.where(flags == 1, np.uint64(delta[1]), np.uint64(0))
    assert (got["d0"] == x0).all() and (got["d1"] == x1).all()  # iknp_test.go:98-113
    seed = oracle.label_from_bytes(drbg("cfg3seed", 16))
    sent = engine.cot_send_pads(ctx, seed, delta, data, wires)
    res = engine.cot_receive_unpad(ctx, seed, flags, sent, got)
    want = np.where(flags.astype(bool), wires["l1"], wires["l0"])
    assert (res == want).all()  # ot_test.go:83-97
    inputs[:, nin_g:] = res.reshape(batch, nin_e)
    out = dc.eval(KEY, g["slab"], inputs=inputs, batch=batch)
    outw = io[:, c.num_inputs:]
    for i in range(batch):
        bits = np.zeros(c.num_outputs, np.uint8)
        is1 = out[i] == outw[i]["l1"]
        is0 = out[i] == outw[i]["l0"]
        assert (is0 ^ is1).all(), "output label of instance %d is neither L0 nor L1" % i
        bits[is1] = 1
        digest = np.packbits(bits, bitorder="little").tobytes()
        assert digest == hashlib.sha256(bytes(x ^ y for x, y in zip(a[i], b[i]))).digest(), "instance %d" % i
        if i == 0:
            assert digest.hex() == "4b2f74579fc7c778745121996f604371a326dc5174f9851706032626668abf2e"
    # instance 0 byte parity with the oracle
    stride = 16 * (c.num_inputs + 1)
    ref = oracle.garble(c.Gates, c.NumWires, c.num_inputs, KEY, rnd[:stride])
    assert (g["slab"][0] == ref["slab"]).all() and g["R"][0] == ref["R"]
    rcv.close(); snd.close(); dc.close(); ctx.close()


def test_config3_device_resident(sha_circ):
    """the same flow with everything in HBM: garble -> input wires gathered for the OT sender -> IKNP extension ->
    COT pads -> the evaluator's labels scattered into its wire array -> eval -> decode; no host round trip between
    the steps (gc_batch_gather_input_wires, gc_iknp_*_dev, gc_cot_*_dev, gc_batch_set_input_range)"""
    c = sha_circ
    batch = 64
    ctx = engine.Context(0)
    dc = engine.DeviceCircuit(ctx, c)
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    nin_g, nin_e = c.Inputs[0], c.Inputs[1]
    n = batch * nin_e
    a = [bytes(range(32))] + [drbg("da%d" % i, 32) for i in range(1, batch)]
    b = [bytes(32 - i for i in range(32))] + [drbg("db%d" % i, 32) for i in range(1, batch)]
    bits = np.zeros((batch, c.num_inputs), np.uint8)
    for i in range(batch):
        bits[i, :nin_g] = bits_of_bytes(a[i])
        bits[i, nin_g:] = bits_of_bytes(b[i])
    rnd = drbg("cfg3dev", 16 * (c.num_inputs + 1) * batch)
    d_rnd = ctx.to_device(rnd)
    d_bits = ctx.to_device(bits)
    flags = np.ascontiguousarray(bits[:, nin_g:]).reshape(-1)
    chunks = (n + 511) // 512
    packed = np.zeros(chunks * 64, np.uint8)
    pk = np.packbits(flags, bitorder="little")
    packed[:len(pk)] = pk
    d_choice = ctx.to_device(packed)
    d_flags = ctx.to_device(flags)
    d_wires = ctx.zeros((n, 32))
    d_u = ctx.zeros(chunks * 8192)
    d_lr = ctx.zeros((n, 16))
    d_ls = ctx.zeros((n, 16))
    d_sent = ctx.zeros((2 * n, 16))
    d_out = ctx.zeros((batch, c.num_outputs))
    d_mis = ctx.zeros(1, np.int32)
    base, delta, k0 = base_setup("cfg3dev")
    rcv, snd = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    seed = oracle.label_from_bytes(drbg("cfg3devseed", 16))
    gb.garble(KEY, d_rnd)
    gb.gather_input_wires(nin_g, nin_e, d_wires)
    rcv.receive_dev(d_choice, n, d_u, d_lr)
    snd.send_dev(d_u, n, d_ls)
    engine.cot_send_pads_dev(ctx, seed, delta, d_ls, d_wires, n, d_sent)
    engine.cot_receive_unpad_dev(ctx, seed, d_flags, d_sent, d_lr, n)
    ev.select_inputs(gb, d_bits)             # the garbler's own input labels (sent in the clear) ...
    ev.set_input_range(nin_g, nin_e, d_lr)   # ... and the evaluator's through the OT
    ev.eval(KEY, gb)
    gb.decode(ev, d_out, d_mis)
    ctx.sync()
    assert int(d_mis.numpy()[0]) == 0
    out = d_out.numpy()
    for i in range(batch):
        digest = np.packbits(out[i], bitorder="little").tobytes()
        assert digest == hashlib.sha256(bytes(x ^ y for x, y in zip(a[i], b[i]))).digest(), "instance %d" % i
    assert np.packbits(out[0], bitorder="little").tobytes().hex() == \
        "4b2f74579fc7c778745121996f604371a326dc5174f9851706032626668abf2e"
    # the OT really delivered the chosen labels: compare with the garbler's wires
    w = d_wires.numpy().view(np.uint64).reshape(n, 4)
    got = d_lr.numpy().view(np.uint64).reshape(n, 2)
    want = np.where(flags[:, None].astype(bool), w[:, 2:], w[:, :2])
    assert (got == want).all()
    rcv.close(); snd.close(); gb.close(); ev.close(); dc.close(); ctx.close()
