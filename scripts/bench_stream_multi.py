#!/usr/bin/env python3
"""Several streams on ONE GPU: N processes of tools/stream_driver (each its own gc_ctx), started together on the same
program.  A single stream's big steps use 32 CUs; config 5 does not shard (replicas only), so replicas are
how a GPU is filled.
usage: bench_stream_multi.py [program] [N ...]     (default: big130, N = 1 2 4)
Prints one JSON line per N: per-process rates and the aggregate over the window in which the timed phases overlap."""
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_stream as bs  # noqa: E402


def run(name, n, path, gates):
    procs = [subprocess.Popen([bs.NATIVE, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(n)]
    res = []
    for p in procs:
        out, err = p.communicate(timeout=240)
        if p.returncode:
            raise RuntimeError("stream_driver failed: %s" % err[-300:])
        res.append(json.loads(out.strip().splitlines()[-1]))
    want = bs.golden_sha(name, bytes(range(32)))
    for r in res:
        assert want is None or r["sha256"] == want, "stream SHA-256 differs from the oracle's"
    out = {"program": name, "streams": n, "gates_per_stream": gates, "sha256_ok": want is not None}
    for side in ("garble", "eval"):
        t0 = min(r[side + "_t0"] for r in res)
        t1 = max(r[side + "_t1"] for r in res)
        overlap = max(r[side + "_t0"] for r in res) < min(r[side + "_t1"] for r in res)
        out[side + "_aggregate_gates_per_s"] = n * gates / (t1 - t0)
        out[side + "_per_stream_gates_per_s"] = [round(gates / r[side + "_s"]) for r in res]
        out[side + "_phases_overlap"] = overlap
    return out


def main():
    args = sys.argv[1:]
    name = args[0] if args else "big130"
    ns = [int(a) for a in args[1:]] or [1, 2, 4]
    steps, prim = bs.PROGRAMS[name]()
    gates = sum(c.NumGates for c, _, _ in steps)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "program.bin")
        bs.write_program(path, bytes(range(32)), bs.stream_rnd(name, len(prim)), prim, steps, 2 if name.startswith("big") else bs.WINDOWS.get(name, 64))
        for n in ns:
            print(json.dumps(run(name, n, path, gates)), flush=True)


if __name__ == "__main__":
    main()
