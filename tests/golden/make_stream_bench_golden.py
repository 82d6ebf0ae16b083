#!/usr/bin/env python3
"""Pins the byte streams of scripts/bench_stream.py's programs with the ORACLE (the CPU restatement of
circuit/stream_garble.go): SHA-256 of the whole serialised stream per program and key size, written to
tests/golden/stream_bench_golden.json.  Run in the build container (no GPU):  python tests/golden/make_stream_bench_golden.py
bench_stream.run_program() refuses a GPU stream whose SHA-256 differs (the bench line's stream.sha256
is a checked value, not a printed one)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from scripts.bench_stream import GOLDEN, PROGRAMS, stream_rnd


def main(names):
    try:
        out = json.load(open(GOLDEN))
    except (OSError, ValueError):
        out = {}
    key = bytes(range(32))
    for name in names:
        t0 = time.time()
        steps, prim = PROGRAMS[name]()
        g = oracle.Stream(key, stream_rnd(name, len(prim)), prim)
        h = hashlib.sha256()
        n = 0
        for c, in_, out_ in steps:
            data = g.garble(c.Gates, c.NumWires, in_, out_)
            h.update(data)
            n += len(data)
        out["%s/key%d" % (name, len(key))] = h.hexdigest()
        print("%-14s %6d steps %12d bytes  %s  (%.1f s)" % (name, len(steps), n, h.hexdigest(), time.time() - t0), flush=True)
    with open(GOLDEN, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main(sys.argv[1:] or ["big", "big130", "uniform512", "uniform4096", "uniform512x64", "mixed", "mixed100", "ssa23", "ed25519like", "ed25519like1"])
