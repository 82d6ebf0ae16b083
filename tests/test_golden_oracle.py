"""The committed golden digests (tests/golden/garble_golden.json) are reproduced by the oracle —
guards the fixture and the oracle against drift.  The GPU suite checks the product against the
same file."""
import hashlib
import importlib.util
import json
import os

import numpy as np

import oracle


def load_mk(golden_dir):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_oracle_reproduces_golden(golden_dir):
    mk = load_mk(golden_dir)
    gold = json.load(open(os.path.join(golden_dir, "garble_golden.json")))
    circs = mk.circuits()
    assert gold["keys"] == mk.KEYS
    for name in ("add64", "comparator64", "synth_allops"):
        for kn, kh in mk.KEYS.items():
            assert mk.golden_for(name, circs[name], bytes.fromhex(kh), mk.COUNTS[name]) == gold["circuits"][name][kn]
    # the two big circuits: one key each keeps the CPU suite quick
    assert mk.golden_for("aes_128", circs["aes_128"], bytes.fromhex(mk.KEYS["aes256"]), 1) == \
        gold["circuits"]["aes_128"]["aes256"][:1]
    assert mk.golden_for("sha256xor", circs["sha256xor"], bytes.fromhex(mk.KEYS["aes128"]), 1) == \
        gold["circuits"]["sha256xor"]["aes128"][:1]


def load_mk2(golden_dir):
    spec = importlib.util.spec_from_file_location("make_golden_stream_ot", os.path.join(golden_dir, "make_golden_stream_ot.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_oracle_reproduces_stream_and_ot_golden(golden_dir):
    """tests/golden/stream_ot_golden.json (streaming garbler bytes, IKNP messages and labels)"""
    mk = load_mk2(golden_dir)
    gold = json.load(open(os.path.join(golden_dir, "stream_ot_golden.json")))
    for b, k in mk.STREAM_CASES:
        assert mk.golden_stream(b, k) == gold["stream"]["%d/%d" % (b, k)]
    for n in mk.IKNP_SIZES:
        assert mk.golden_iknp(n) == gold["iknp"][str(n)]
