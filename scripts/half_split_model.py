#!/usr/bin/env python3
"""VERDICT r3 item 5, priced before building it: split every AND into its two half-gate hashes as separately schedulable items
of the flattened plan — H(a, j0) needs only operand a (circuit/eval.go:62-75, garble.go:362-376), so it may run in ANY hash
phase after a exists, parking its 16 bytes (the garbler: H(a0) and H(a0)^H(a1), 32 bytes) in LDS slots until the phase that
has b.  What a pass can gain is bounded by two things this script computes from the plan of the circuit (no GPU):

  * wave items.  A hash phase costs ceil(waves / 4) wave-long AES items on its busiest SIMD (the kernels sit on the
    SIMD issue bound of the T-table AES with the granularity of whole waves); moving halves between phases only helps by
    filling partial sets of four waves — the packing bound is ceil(all waves / 4);
  * live labels.  A parked half is a live label from its phase to the combine; the tile's LDS holds what is left beside the
    64 KiB table and the stage buffers: 1 067 labels per instance at TI = 4.  aes_128 peaks at 1 052 already.

Greedy packing under the live-label budget: for every phase whose last set of four waves is partial, move just enough early
halves to earlier phases that have room in THEIR last set (latest possible phase first: shortest parking), never beyond the
budget.  Prints the per-phase table and the projected pass times (hash part = 78 % / 86 % of an eval / garble pass).

usage: scripts/half_split_model.py [circuit.gcf] [TI]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from mpc_amd import parse_file
from mpc_amd.circuit import AND, INV, OR, XNOR, XOR

BUDGET = {4: 1067, 2: 2100, 1: 4200}  # live labels per instance beside table + stage buffers (plan.h: kFlatLdsBytes)


def live_labels(path):
    """live labels after every step of the flattened schedule (the planner's own account, GC_PLAN_DEBUG)"""
    code = ("import sys; sys.path.insert(0, %r)\nimport numpy as np\nfrom mpc_amd import engine, parse_file\n"
            "from mpc_amd.circuit import GATE\nc = parse_file(%r)\n"
            "engine.Plan(np.ascontiguousarray(c.Gates, dtype=GATE), c.NumWires, c.num_inputs, c.num_outputs)\n" % (ROOT, path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, GC_PLAN_DEBUG="1"))
    live, kind = [], []
    for ln in r.stderr.splitlines():
        if ln.startswith("[plan] step"):
            f = ln.split()
            kind.append(f[3])
            live.append(int(f[-1].split("=")[1]))
    return kind, live


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "aes_128.gcf")
    TI = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    c = parse_file(path)
    g = c.Gates
    depth = np.zeros(c.NumWires, np.int32)
    ands = []  # (phase, depth of the earlier operand)
    others = {}  # phase -> [n_or, n_inv]
    for i0, i1, o, op in zip(g["in0"].tolist(), g["in1"].tolist(), g["out"].tolist(), g["op"].tolist()):
        if op in (XOR, XNOR):
            depth[o] = max(depth[i0], depth[i1])
        elif op == INV:
            depth[o] = depth[i0] + 1
            others.setdefault(int(depth[o]), [0, 0])[1] += 1
        else:
            da, db = int(depth[i0]), int(depth[i1])
            depth[o] = max(da, db) + 1
            if op == AND:
                ands.append((int(depth[o]), min(da, db), da != db))
            else:
                others.setdefault(int(depth[o]), [0, 0])[0] += 1
    nph = int(depth.max())
    kind, live = live_labels(path)
    # live labels while hash phase p runs = after the XOR round(s) in front of it; the k-th "hash" step is phase k + 1
    hash_live, last = [], 0
    for k, lv in zip(kind, live):
        if k == "hash":
            hash_live.append(max(last, lv))
        last = lv
    while len(hash_live) < nph:
        hash_live.append(last)
    budget = BUDGET[TI]
    print("%s: %d ANDs in %d hash phases, TI = %d, live-label budget %d per instance, peak now %d" % (
        os.path.basename(path), len(ands), nph, TI, budget, max(live)))
    for side, lanes_and, lanes_half, park in (("eval", 2, 1, 1), ("garble", 4, 2, 2)):
        lanes = [0] * (nph + 1)
        movable = {p: [] for p in range(1, nph + 1)}  # phase -> [earliest phase the early half may run in]
        for p, dmin, split in ands:
            lanes[p] += lanes_and * TI
            if split:
                movable[p].append(dmin + 1)
        for p, (n_or, n_inv) in others.items():
            lanes[p] += ((4 if side == "garble" else 1) * n_or + (2 if side == "garble" else 1) * n_inv) * TI
        items = lambda L: -(-(-(-L // 64)) // 4)
        before = [items(L) for L in lanes]
        extra = [0] * (nph + 2)  # parked labels alive while phase q runs
        L = list(lanes)
        moved = 0
        for p in range(2, nph + 1):
            r = L[p] % 256
            if r == 0 or not movable[p]:
                continue
            need = -(-r // (lanes_half * TI))  # halves to move away so that the partial set of four waves disappears
            if need > len(movable[p]):
                continue
            plan, ok = [], True
            cand = sorted(movable[p], reverse=True)  # latest availability first: they have the shortest possible parking
            extra_try, L_try = list(extra), list(L)
            for lo in cand[:need]:
                placed = False
                for q in range(p - 1, lo - 1, -1):
                    room = (256 - L_try[q] % 256) % 256
                    if room < lanes_half * TI:
                        continue
                    if any(hash_live[t - 1] + extra_try[t] + park > budget for t in range(q, p + 1)):
                        continue
                    L_try[q] += lanes_half * TI
                    for t in range(q, p + 1):
                        extra_try[t] += park
                    placed = True
                    break
                if not placed:
                    ok = False
                    break
            if ok:
                L_try[p] -= need * lanes_half * TI
                L, extra = L_try, extra_try
                moved += need
        after = [items(x) for x in L]
        waves = sum(-(-x // 64) for x in lanes)
        print("\n%s pass: %d lanes = %d waves; wave items on the busiest SIMD: %d now, %d with %d halves moved under the budget, "
              "packing bound %d (no budget: every phase a multiple of four waves)" % (
                  side, sum(lanes), waves, sum(before), sum(after), moved, -(-waves // 4)))
        share = 0.78 if side == "eval" else 0.86
        now = {"eval": 0.382, "garble": 0.622}[side] if "aes_128" in path and TI == 4 else None
        gain = 1 - sum(after) / max(sum(before), 1)
        print("  hash part = %.0f %% of the pass -> at best %.1f %% of the pass%s; parked labels at the peak: +%d (live %d of %d)" % (
            100 * share, 100 * share * gain, (" = %.3f -> %.3f ms" % (now, now * (1 - share * gain))) if now else "",
            max(extra), max(hash_live[t - 1] + extra[t] for t in range(1, nph + 1)), budget))
        print("  phase: lanes now -> split | items now -> split | live labels now +parked")
        for p in range(1, nph + 1):
            if lanes[p] != L[p] or p <= 12:
                print("  %5d: %6d -> %6d | %2d -> %2d | %5d +%d" % (p, lanes[p], L[p], before[p], after[p], hash_live[p - 1], extra[p]))


if __name__ == "__main__":
    main()
