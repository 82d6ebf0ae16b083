#!/usr/bin/env python3
"""The level-launch row of bench.py by itself (schedule 0: one launch per dependency level): aes_128 x 1024 under the 32-byte
key, once per form of the hash workgroups — GC_LEVEL_CLASSIC=1 (split kernels on the classic tables) and the default (the
production AES core, a shared dual-table image per workgroup) — each in a process of its own (the switch is read once)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import bench
    from mpc_amd import engine, parse_file
    circ = parse_file(os.path.join(ROOT, "tests", "golden", "aes_128.gcf"))
    ctx = engine.Context(0)
    row = bench.level_launch_row(int(os.environ.get("LL_BATCH", "1024")), circ, bytes(range(32)), ctx)
    ctx.close()
    print(json.dumps(row))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for form, env in (("classic", {"GC_LEVEL_CLASSIC": "1"}), ("dual", {})):
            e = dict(os.environ)
            e.update(env)
            out = subprocess.run([sys.executable, __file__, "one"], env=e, capture_output=True, text=True)
            line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]
            print(form, line)
