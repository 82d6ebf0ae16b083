#!/usr/bin/env python3
"""Generates tests/golden/garble_golden.json with the CPU oracle (oracle/): SHA-256 digests of
R || slab || output-wire labels (Go memory layout) and of the evaluated wire labels, for seeded
random streams.  The product (HIP path) must reproduce these byte-for-byte; the CPU suite checks
that the oracle still does.  Deterministic: re-running must not change the file."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from mpc_amd.circuit import comparator64, parse_file, synthetic_levelised  # noqa: E402
from tests.util import drbg  # noqa: E402


def circuits():
    return {
        "aes_128": parse_file(os.path.join(HERE, "aes_128.gcf")),
        "sha256xor": parse_file(os.path.join(HERE, "sha256xor.gcf")),
        "add64": parse_file(os.path.join(HERE, "add64.gcf")),
        "comparator64": comparator64(),
        "synth_allops": synthetic_levelised(10, 48, 0.3, seed=5, ninputs=40, or_frac=0.1, inv_frac=0.1,
                                            xnor_frac=0.1),
    }


def instance_streams(name, c, i):
    rnd = drbg("golden/%s/rnd/%d" % (name, i), 16 * (c.num_inputs + 1))
    bits = np.frombuffer(drbg("golden/%s/in/%d" % (name, i), c.num_inputs), np.uint8) & 1
    return rnd, bits


def golden_for(name, c, key, n):
    out = []
    for i in range(n):
        rnd, bits = instance_streams(name, c, i)
        g = oracle.garble(c.Gates, c.NumWires, c.num_inputs, key, rnd)
        h = hashlib.sha256()
        h.update(g["R"].tobytes())
        h.update(g["slab"].tobytes())
        h.update(np.ascontiguousarray(g["wires"]["l0"][c.NumWires - c.num_outputs:]).tobytes())
        w = np.zeros(c.NumWires, oracle.LABEL)
        w[: c.num_inputs] = np.where(bits.astype(bool), g["wires"]["l1"][: c.num_inputs],
                                     g["wires"]["l0"][: c.num_inputs])
        oracle.eval_(c.Gates, c.NumWires, key, w, g["slab"])
        out.append({"garble": h.hexdigest(), "eval": hashlib.sha256(w.tobytes()).hexdigest()})
    return out


KEYS = {"aes256": bytes(range(32)).hex(), "aes128": b"0123456789abcdef".hex(), "aes192": bytes(range(7, 31)).hex()}
COUNTS = {"aes_128": 3, "sha256xor": 2, "add64": 4, "comparator64": 4, "synth_allops": 4}

if __name__ == "__main__":
    res = {"keys": KEYS, "circuits": {}}
    for name, c in circuits().items():
        res["circuits"][name] = {kn: golden_for(name, c, bytes.fromhex(kh), COUNTS[name]) for kn, kh in KEYS.items()}
        print(name, "done")
    with open(os.path.join(HERE, "garble_golden.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
