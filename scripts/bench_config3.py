#!/usr/bin/env python3
"""BASELINE config 3 end to end, device-resident: sha256xor x 256 instances — garble -> the evaluator's 256 input wires per
instance gathered for the OT sender -> IKNP extension of 65 536 OTs (receiver + sender) -> COT pads / unpad -> eval ->
decode, with no host round trip between the steps.  Checks every digest against hashlib (instance 0 is the reference's
sha2pc_test.go:124 vector) and prints one JSON line with the wall time per pipeline pass."""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mpc_amd import engine, parse_file
from mpc_amd.circuit import LABEL, WIRE


def drbg(tag, n):
    return hashlib.shake_256(("cfg3-bench/" + tag).encode()).digest(n)


def lab(raw):  # Label.SetData (ot/label.go:112-115): big-endian D0 || D1
    return (int.from_bytes(raw[:8], "big"), int.from_bytes(raw[8:16], "big"))


def bits_of_bytes(b):
    return np.unpackbits(np.frombuffer(b, np.uint8), bitorder="little")


def run(batch=256, reps=20, key=bytes(range(32))):
    c = parse_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "sha256xor.gcf"))
    ctx = engine.Context(0)
    dc = engine.DeviceCircuit(ctx, c)
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    nin_g, nin_e = c.Inputs[0], c.Inputs[1]
    n = batch * nin_e
    a = [bytes(range(32))] + [drbg("a%d" % i, 32) for i in range(1, batch)]
    b = [bytes(32 - i for i in range(32))] + [drbg("b%d" % i, 32) for i in range(1, batch)]
    bits = np.zeros((batch, c.num_inputs), np.uint8)
    for i in range(batch):
        bits[i, :nin_g] = bits_of_bytes(a[i])
        bits[i, nin_g:] = bits_of_bytes(b[i])
    d_rnd = ctx.to_device(drbg("rnd", 16 * (c.num_inputs + 1) * batch))
    d_bits = ctx.to_device(bits)
    flags = np.ascontiguousarray(bits[:, nin_g:]).reshape(-1)
    chunks = (n + 511) // 512
    packed = np.zeros(chunks * 64, np.uint8)
    pk = np.packbits(flags, bitorder="little")
    packed[:len(pk)] = pk
    d_choice, d_flags = ctx.to_device(packed), ctx.to_device(flags)
    z = lambda *shape: ctx.zeros(shape)
    d_wires, d_u, d_lr, d_ls, d_sent = z(n, 32), z(chunks * 8192), z(n, 16), z(n, 16), z(2 * n, 16)
    d_out = z(batch, c.num_outputs)
    d_mis = ctx.zeros(1, np.int32)
    base = np.zeros(128, WIRE)
    for i in range(128):
        base[i]["l0"], base[i]["l1"] = lab(drbg("l0/%d" % i, 16)), lab(drbg("l1/%d" % i, 16))
    delta = lab(drbg("delta", 16))
    k0 = np.zeros(128, LABEL)
    for i in range(128):  # Delta.Bit(i): bit i of D0 for i < 64, of D1 above (ot/label.go:129-141; D0 is d[0])
        bit = (delta[0] >> i) & 1 if i < 64 else (delta[1] >> (i - 64)) & 1
        k0[i] = base[i]["l1"] if bit else base[i]["l0"]
    seed = lab(drbg("seed", 16))

    def one_pass():
        # fresh OT state per pass: the column streams of an IKNP pair advance with every call
        rcv, snd = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
        ctx.sync()
        t0 = time.perf_counter()
        gb.garble(key, d_rnd)
        gb.gather_input_wires(nin_g, nin_e, d_wires)
        rcv.receive_dev(d_choice, n, d_u, d_lr)
        snd.send_dev(d_u, n, d_ls)
        engine.cot_send_pads_dev(ctx, seed, delta, d_ls, d_wires, n, d_sent)
        engine.cot_receive_unpad_dev(ctx, seed, d_flags, d_sent, d_lr, n)
        ev.select_inputs(gb, d_bits)
        ev.set_input_range(nin_g, nin_e, d_lr)
        ev.eval(key, gb)
        gb.decode(ev, d_out, d_mis)
        ctx.sync()
        dt = time.perf_counter() - t0
        rcv.close()
        snd.close()
        return dt

    one_pass()
    times = [one_pass() for _ in range(reps)]
    assert int(d_mis.numpy()[0]) == 0
    out = d_out.numpy()
    for i in range(batch):
        digest = np.packbits(out[i], bitorder="little").tobytes()
        assert digest == hashlib.sha256(bytes(x ^ y for x, y in zip(a[i], b[i]))).digest(), "instance %d" % i
    assert np.packbits(out[0], bitorder="little").tobytes().hex() == \
        "4b2f74579fc7c778745121996f604371a326dc5174f9851706032626668abf2e"
    t = float(np.median(times))
    res = {"workload": "sha256xor x %d, %d OTs, device-resident pipeline" % (batch, n), "ms_per_pass": t * 1e3,
           "and_gates_per_s": dc.info.n_and * batch / t, "instances_per_s": batch / t, "garble_ms": gb.last_ms,
           "eval_ms": ev.last_ms, "digests_ok": True}
    gb.close(); ev.close(); dc.close(); ctx.close()
    return res


if __name__ == "__main__":
    print(json.dumps(run()))
