#!/usr/bin/env python3
"""VERDICT r5 item 5, priced from the plan (no GPU): slack-aware levelling of hashed gates over the hash phases of the
flattened schedule.  A hash phase costs its busiest SIMD ceil(waves / 4) wave-long AES items (whole waves, 16 per
workgroup, 4 SIMDs), with the last partial set of four waves column-sliced when it is short (split_hash_lanes: quarter-cost
waves on every SIMD).  For every hashed gate: its ASAP phase (the production schedule's) and its ALAP phase (latest phase its
consumers allow); a gate may sit anywhere in between WITHOUT changing the number of phases.  Printed: items per pass today,
the perfect-packing bound, and what a greedy list scheduler over the slack reaches — whole gates, eval (2 lanes per AND and
instance) and garble (4).

usage: scripts/phase_balance_model.py [circuit.gcf] [TI]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from mpc_amd import parse_file
from mpc_amd.circuit import AND, INV, OR, XNOR, XOR

TAIL_MAX = {"eval": 192, "garble": 128}


def items(lanes, side):
    """AES items on the busiest SIMD for a phase of `lanes` wide-form hash lanes (fused_flat_kernels.hip: split_hash_lanes)"""
    if lanes == 0:
        return 0.0
    full, rem = divmod(lanes, 1024)
    cost = 4.0 * full
    if rem:
        small = rem <= 256
        k, r = (0, rem) if small else (rem >> 8, rem & 255)
        fits = k == 0 or (r <= TAIL_MAX[side] and ((r + 15) >> 4) + 4 * k <= 16)
        if fits:  # r blocks column-sliced: ceil(4 r / 64 / 4) quarter-cost waves per SIMD
            cost += k + (-(-r // 64)) * 0.25 * (2.2 / 3.3 * 4)  # a quarter-wave's AES is ~2.2k of a wide wave's 3.3k cycles, 4 lanes per block
        else:
            cost += k + 1
    return cost


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "aes_128.gcf")
    TI = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    c = parse_file(path)
    g = c.Gates
    n = len(g)
    in0, in1, out, op = g["in0"].tolist(), g["in1"].tolist(), g["out"].tolist(), g["op"].tolist()
    depth = np.zeros(c.NumWires, np.int32)      # non-free depth of every wire (ASAP phase of the gate that wrote it)
    writer = np.full(c.NumWires, -1, np.int64)
    gphase = np.zeros(n, np.int32)
    for i in range(n):
        d = max(depth[in0[i]], depth[in1[i]] if op[i] != INV else 0)
        if op[i] not in (XOR, XNOR):
            d += 1
        depth[out[i]] = d
        gphase[i] = d
        writer[out[i]] = i
    nph = int(depth.max())
    # ALAP: latest phase of every hashed gate such that every hashed consumer (through free gates) still runs one phase later;
    # circuit outputs may be as late as the last phase
    late_wire = np.full(c.NumWires, nph, np.int32)  # latest phase in which the wire's value must exist (produced in a phase <= this)
    alap = np.zeros(n, np.int32)
    for i in range(n - 1, -1, -1):
        o = out[i]
        if op[i] in (XOR, XNOR):
            need = late_wire[o]
        else:
            alap[i] = late_wire[o]
            need = alap[i] - 1
        late_wire[in0[i]] = min(late_wire[in0[i]], need)
        if op[i] != INV:
            late_wire[in1[i]] = min(late_wire[in1[i]], need)
    hashed = [i for i in range(n) if op[i] not in (XOR, XNOR)]
    slack = np.array([alap[i] - gphase[i] for i in hashed])
    print("%s: %d hashed gates, %d hash phases, TI = %d; slack (ALAP - ASAP phase): 0 for %d gates, 1-2 for %d, >= 3 for %d" % (
        os.path.basename(path), len(hashed), nph, TI, int((slack == 0).sum()), int(((slack > 0) & (slack < 3)).sum()), int((slack >= 3).sum())))
    for side, lanes_per in (("eval", {AND: 2, OR: 1, INV: 1}), ("garble", {AND: 4, OR: 4, INV: 2})):
        def phase_lanes(assign):
            L = np.zeros(nph + 1, np.int64)
            for i, p in zip(hashed, assign):
                L[p] += lanes_per[op[i]] * TI
            return L
        asap = phase_lanes([gphase[i] for i in hashed])
        now = sum(items(int(x), side) for x in asap)
        total = int(asap.sum())
        bound = total / 256.0  # every SIMD busy with whole waves all the time
        # greedy list scheduling over the slack: phases in order, a gate whose ASAP phase has come is placed now if the phase
        # still has room in whole sets of four waves, else deferred while its slack allows
        cap_sets = np.ceil(asap / 256.0)
        assign = {}
        pending = []
        by_asap = {}
        for i in hashed:
            by_asap.setdefault(int(gphase[i]), []).append(i)
        target = -(-total // (256 * nph)) * 256  # lanes per phase for an even spread, whole sets of four waves
        for p in range(1, nph + 1):
            pending += by_asap.get(p, [])
            pending.sort(key=lambda i: alap[i])
            used, keep = 0, []
            for i in pending:
                w = lanes_per[op[i]] * TI
                if alap[i] <= p or used + w <= max(target, 256):
                    assign[i] = p
                    used += w
                else:
                    keep.append(i)
            pending = keep
        assert not pending
        # (a deferred gate's consumers must move too: this greedy only defers gates with slack, so dependencies hold by ALAP)
        bal = phase_lanes([assign[i] for i in hashed])
        greedy = sum(items(int(x), side) for x in bal)
        waves = np.ceil(asap / 64.0)
        print("  %-6s lanes per phase: min %d median %d max %d; waves per phase %.1f avg; items on the busiest SIMD per pass: today %.1f, "
              "greedy over the slack %.1f (%.1f %%), perfect packing %.1f (%.1f %%)" % (
                  side, asap[1:].min(), int(np.median(asap[1:])), asap.max(), waves[1:].mean(), now, greedy, 100 * (now - greedy) / now,
                  bound, 100 * (now - bound) / now))


if __name__ == "__main__":
    main()
