#!/usr/bin/env python3
"""Generates tests/golden/stream_ot_golden.json with the CPU oracle (oracle/): SHA-256 digests of
  * the streaming garbler's byte stream (circuit/stream_garble.go wire format) and the labels of the program's output
    wires, for the three-step program of tests/test_oracle_stream.py at 16-bit and 32-bit wire ids,
  * the IKNP extension (ot/iknp.go): the u-matrix message, the receiver's labels and the sender's labels, with base-OT
    seeds and choice bits from the test DRBG.
The product (HIP path) must reproduce these byte for byte (tests/test_gpu_stream.py, tests/test_gpu_ot.py); the CPU suite
checks that the oracle still does.  Deterministic: re-running must not change the file."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from mpc_amd.circuit import LABEL, WIRE  # noqa: E402
from tests.test_oracle_stream import make_program  # noqa: E402
from tests.util import drbg  # noqa: E402

STREAM_CASES = [(0, 32), (0x20000, 16), (70000, 24)]
IKNP_SIZES = [1, 700, 5000]


def stream_inputs(base, keylen):
    steps, prim = make_program(base)
    return steps, prim, drbg("gold-skey", keylen), drbg("gold-srnd%d" % base, 16 * (len(prim) + 1))


def stream_digest(garbler, steps):
    """garbler: oracle.Stream or engine.Stream (same interface)"""
    h = hashlib.sha256()
    for c, in_, out_ in steps:
        h.update(garbler.garble(c.Gates, c.NumWires, in_, out_))
    outs = hashlib.sha256()
    for c, in_, out_ in steps:
        for o in out_:
            w = garbler.get(o)
            outs.update(np.array([w["l0"]["d0"], w["l0"]["d1"]], np.uint64).tobytes())
    return {"stream": h.hexdigest(), "out_labels": outs.hexdigest()}


def labels(seed, n):
    raw = drbg(seed, 16 * n)
    out = np.zeros(n, LABEL)
    for i in range(n):
        out[i] = oracle.label_from_bytes(raw[16 * i:16 * i + 16])
    return out


def iknp_inputs(n):
    base = np.zeros(128, WIRE)
    base["l0"] = labels("gold-ik-l0", 128)
    base["l1"] = labels("gold-ik-l1", 128)
    delta = oracle.label_from_bytes(drbg("gold-ik-delta", 16))
    k0 = np.zeros(128, LABEL)
    for i in range(128):
        k0[i] = base[i]["l1"] if oracle.label_bit(delta, i) else base[i]["l0"]
    b = (np.frombuffer(drbg("gold-ik-b%d" % n, n), np.uint8) & 1).astype(np.uint8)
    return base, delta, k0, b


def iknp_digest(u, got, sent):
    return {"u": hashlib.sha256(bytes(u)).hexdigest(),
            "receiver": hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest(),
            "sender": hashlib.sha256(np.ascontiguousarray(sent).tobytes()).hexdigest()}


def golden_stream(base, keylen):
    steps, prim, key, rnd = stream_inputs(base, keylen)
    return stream_digest(oracle.Stream(key, rnd, prim), steps)


def golden_iknp(n):
    base, delta, k0, b = iknp_inputs(n)
    rcv, snd = oracle.IKNPReceiver(base), oracle.IKNPSender(delta, k0)
    u, got = rcv.receive(b)
    return iknp_digest(u, got, snd.send(u, n))


if __name__ == "__main__":
    res = {"stream": {"%d/%d" % (b, k): golden_stream(b, k) for b, k in STREAM_CASES},
           "iknp": {str(n): golden_iknp(n) for n in IKNP_SIZES}}
    with open(os.path.join(HERE, "stream_ot_golden.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("written")
