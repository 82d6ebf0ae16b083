#!/usr/bin/env python3
"""PCIe-inclusive rate of the literal drop-in calls (gc_garble / gc_eval with HOST buffers): every call moves
the caller's random stream in and R, the input/output wires and the table slab out (garble), the slab and input labels in
and the output labels out (eval).  Reported next to the HBM-resident rate of bench.py, never instead of it.

Three figures per batch size:
  pageable   the caller's buffers are ordinary heap memory (what an unmodified Go caller passes)
  pinned     the buffers come from gc_host_alloc (the shim's scratch pool): direct DMA, chunk-pipelined
  duplex     pinned, garbler and evaluator as two threads with their own gc_ctx: garble of batch k+1 (tables device ->
             host) overlaps eval of batch k (tables host -> device) — the two directions of the PCIe link; in the real
             protocol the two passes run on two machines and each party sees its own direction only
"""
import ctypes as C
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mpc_amd import engine, parse_file
from mpc_amd.circuit import LABEL, WIRE

p = lambda a: a.ctypes.data_as(C.c_void_p)


class Party:
    def __init__(self, c, batch, pinned, nslabs=1):
        self.c, self.batch = c, batch
        self.ctx = engine.Context(0)
        self.dc = engine.DeviceCircuit(self.ctx, c)
        rows, nin, nout = self.dc.info.slab_rows, c.num_inputs, c.num_outputs
        self.rows, self.nin, self.nout = rows, nin, nout
        self._keep = []

        def arr(shape, dtype):
            if not pinned:
                return np.zeros(shape, dtype)
            pa = engine.PinnedArray(shape, dtype)
            pa.a[...] = np.zeros((), dtype)
            self._keep.append(pa)
            return pa.a

        self.R = arr((batch,), LABEL)
        self.slabs = [arr((batch, rows), LABEL) for _ in range(nslabs)]
        self.io = arr((batch, nin + nout), WIRE)
        self.inputs = arr((batch, nin), LABEL)
        self.outl = arr((batch, nout), LABEL)

    def close(self):
        self.dc.close()
        self.ctx.close()
        for pa in self._keep:
            pa.close()


def run(batch=1024, reps=4, key=bytes(range(32)), circuit=None):
    c = parse_file(circuit or os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "aes_128.gcf"))
    L = engine.lib()
    nin = c.num_inputs
    rnd = np.frombuffer(np.random.default_rng(1).bytes(batch * 16 * (nin + 1)), np.uint8).copy()
    k = np.frombuffer(key, np.uint8).copy()
    res = {"batch": batch}

    def garble(P, slab):
        rc = L.gc_garble(P.dc.h, p(k), len(k), p(rnd), len(rnd), batch, p(P.R), None, p(P.io), p(slab))
        assert rc == 0, rc

    def evaluate(P, slab, inputs):
        rc = L.gc_eval(P.dc.h, p(k), len(k), batch, None, p(inputs), p(slab), P.rows, p(P.outl))
        assert rc == 0, rc

    bits = c.compute_bits(np.zeros(nin, np.uint8))[c.NumWires - c.num_outputs:].astype(bool)  # zero inputs

    for mode in ("pageable", "pinned"):
        P = Party(c, batch, mode == "pinned")
        tg, te = [], []
        for r in range(reps + 1):
            t0 = time.perf_counter()
            garble(P, P.slabs[0])
            t1 = time.perf_counter()
            P.inputs[...] = P.io[:, :nin]["l0"]  # all-zero inputs: the L0 labels
            t2 = time.perf_counter()
            evaluate(P, P.slabs[0], P.inputs)
            t3 = time.perf_counter()
            if r:  # the first round warms allocations
                tg.append(t1 - t0)
                te.append(t3 - t2)
        want = np.where(bits[None, :], P.io[:, nin:]["l1"], P.io[:, nin:]["l0"])
        assert (P.outl == want).all()
        g, e = min(tg), min(te)
        n_and = P.dc.info.n_and * batch
        nbytes = P.slabs[0].nbytes
        res[mode] = {"garble_ms": g * 1e3, "eval_ms": e * 1e3, "and_gates_per_s": n_and / (g + e),
                     "garble_GBs_out": (nbytes + P.io.nbytes) / g / 1e9, "eval_GBs_in": (nbytes + P.inputs.nbytes) / e / 1e9}
        res["slab_MB"] = nbytes / 1e6
        P.close()

    # duplex: garbler thread and evaluator thread, two slabs in flight
    G, E = Party(c, batch, True, nslabs=2), Party(c, batch, True, nslabs=0)
    nrep = 2 * reps + 2
    full = [threading.Semaphore(0), threading.Semaphore(0)]
    free = [threading.Semaphore(1), threading.Semaphore(1)]
    inputs = [np.zeros((batch, nin), LABEL), np.zeros((batch, nin), LABEL)]
    ok = [True]
    stamps = []

    def garbler():
        for r in range(nrep):
            s = r & 1
            free[s].acquire()
            garble(G, G.slabs[s])
            inputs[s][...] = G.io[:, :nin]["l0"]
            full[s].release()

    def evaluator():
        for r in range(nrep):
            s = r & 1
            full[s].acquire()
            E.inputs[...] = inputs[s]
            evaluate(E, G.slabs[s], E.inputs)
            stamps.append(time.perf_counter())
            free[s].release()

    tg_, te_ = threading.Thread(target=garbler), threading.Thread(target=evaluator)
    tg_.start(); te_.start(); tg_.join(); te_.join()
    want = np.where(bits[None, :], G.io[:, nin:]["l1"], G.io[:, nin:]["l0"])
    ok[0] = bool((E.outl == want).all())
    assert ok[0]
    per = (stamps[-1] - stamps[1]) / (len(stamps) - 2)  # steady state: skip the warm-up batch
    res["duplex"] = {"ms_per_batch": per * 1e3, "and_gates_per_s": G.dc.info.n_and * batch / per,
                     "GBs_each_way": G.slabs[0].nbytes / per / 1e9}
    G.close(); E.close()
    res["link"] = link_probe()
    return res


def link_probe(n=256 << 20):
    """what the box's host link does for plain pinned-memory DMA (each direction alone, both at once): the bound the
    figures above are to be read against.  Pinned host memory from gc_host_alloc, device memory from gc_dev_alloc, the
    copies are gc_dev_download / gc_dev_upload on two contexts (two streams) driven by two threads."""
    import threading

    c1, c2 = engine.Context(0), engine.Context(0)
    h_out, h_in = engine.PinnedArray(n, np.uint8), engine.PinnedArray(n, np.uint8)
    h_in.a[:] = 3
    d_a, d_b = c1.zeros(n), c2.empty(n)
    L = engine.lib()

    def d2h():
        L.gc_dev_download(c1.h, p(h_out.a), C.c_void_p(d_a.ptr), n)

    def h2d():
        L.gc_dev_upload(c2.h, C.c_void_p(d_b.ptr), p(h_in.a), n)

    def both():
        t = threading.Thread(target=d2h)
        t.start()
        h2d()
        t.join()

    def timed(fn, reps=4):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    out = {"d2h_GBs": n / timed(d2h) / 1e9, "h2d_GBs": n / timed(h2d) / 1e9,
           "both_at_once_GBs_each": n / timed(both) / 1e9}
    d_a.close(); d_b.close(); h_out.close(); h_in.close(); c1.close(); c2.close()
    return out


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)))
