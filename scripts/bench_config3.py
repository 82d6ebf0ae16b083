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
import torch

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
    d_rnd = torch.frombuffer(bytearray(drbg("rnd", 16 * (c.num_inputs + 1) * batch)), dtype=torch.uint8).cuda()
    d_bits = torch.from_numpy(bits.copy()).cuda()
    flags = np.ascontiguousarray(bits[:, nin_g:]).reshape(-1)
    chunks = (n + 511) // 512
    packed = np.zeros(chunks * 64, np.uint8)
    pk = np.packbits(flags, bitorder="little")
    packed[:len(pk)] = pk
    d_choice, d_flags = torch.from_numpy(packed).cuda(), torch.from_numpy(flags.copy()).cuda()
    z = lambda *shape: torch.zeros(shape, dtype=torch.uint8, device="cuda")
    d_wires, d_u, d_lr, d_ls, d_sent = z(n, 32), z(chunks * 8192), z(n, 16), z(n, 16), z(2 * n, 16)
    d_out = z(batch, c.num_outputs)
    d_mis = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
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
        gb.garble(key, d_rnd.data_ptr())
        gb.gather_input_wires(nin_g, nin_e, d_wires.data_ptr())
        rcv.receive_dev(d_choice.data_ptr(), n, d_u.data_ptr(), d_lr.data_ptr())
        snd.send_dev(d_u.data_ptr(), n, d_ls.data_ptr())
        engine.cot_send_pads_dev(ctx, seed, delta, d_ls.data_ptr(), d_wires.data_ptr(), n, d_sent.data_ptr())
        engine.cot_receive_unpad_dev(ctx, seed, d_flags.data_ptr(), d_sent.data_ptr(), d_lr.data_ptr(), n)
        ev.select_inputs(gb, d_bits.data_ptr())
        ev.set_input_range(nin_g, nin_e, d_lr.data_ptr())
        ev.eval(key, gb)
        gb.decode(ev, d_out.data_ptr(), d_mis.data_ptr())
        ctx.sync()
        dt = time.perf_counter() - t0
        rcv.close()
        snd.close()
        return dt

    one_pass()
    times = [one_pass() for _ in range(reps)]
    assert int(d_mis.cpu()[0]) == 0
    out = d_out.cpu().numpy()
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
