#!/usr/bin/env python3
"""Rows that a lone wave's instruction stream bounds, for scripts/r06c_tf_probe.sh: buildANDChain(10000) x 1024 (16-byte key, as in
bench.py), add64 x 256 and sha256xor x 256 (32-byte key), garble_ms / eval_ms by HIP events."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_amd import engine, parse_file
from mpc_amd.circuit import and_chain
from scripts.sweep_synthetic import run as sweep_run

ctx = engine.Context(0)
g = os.path.join(ROOT, "tests", "golden")
for batch, key, circ in ((1024, bytes(range(16)), and_chain(10000)), (256, bytes(range(32)), parse_file(os.path.join(g, "add64.gcf"))),
                         (256, bytes(range(32)), parse_file(os.path.join(g, "sha256xor.gcf"))),
                         (1024, bytes(range(32)), parse_file(os.path.join(g, "aes_128.gcf")))):
    r = sweep_run(batch, 131072, key, ctx=ctx, cases=[], chain=0, circuits=[circ])[0]
    print(json.dumps({k: r[k] for k in ("circuit", "garble_ms", "eval_ms", "and_gates_per_s", "outputs_ok", "hash_phases") if k in r}
                     | {"batch": batch}))
ctx.close()
