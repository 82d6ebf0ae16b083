"""Extended differential fuzz run (not collected by pytest): N (default 300) random circuits x schedules 1 and 2 against
the oracle, batches from 1 to 16 500 instances (tiles of 1 to 64 instances per workgroup).
Run on a GPU box: python tests/ext_fuzz.py [N]"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("GC_FUZZ_DEFAULT_PLANNER"):  # the product's default instead: plans in the background, unplanned chains step by step
    os.environ.pop("GC_STREAM_FUSE_EAGER", None)
else:
    os.environ.setdefault("GC_STREAM_FUSE_EAGER", "1")  # every chain of the streamed cases on its merged plan (as tests/conftest.py)
import oracle
from mpc_amd import engine
from tests.test_gpu_fuzz import random_circuit, xor_tree, KEY
from tests.test_gpu_garble_eval import check_garble_eval
ctx = engine.Context(0)
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for seed in range(N):
    rng = np.random.default_rng(5000 + seed)
    ninputs = int(rng.integers(2, 80))
    ngates = int(rng.integers(1, 3000))
    c = random_circuit(rng, ninputs, ngates, p_xor=float(rng.choice([0.0, 0.3, 0.6, 0.8, 0.9, 0.97, 1.0])),
                       reuse=float(rng.choice([0.0, 0.02, 0.1, 0.3])), nout=int(rng.integers(1, 40)))
    batch = int(rng.choice([1, 2, 5, 64, 130, 520, 1030, 2100, 4100, 16500]))
    sample = None if batch <= 130 else sorted(set(list(range(0, batch, 97)) + [batch - 1, batch - 2, 1]))
    try:
        for schedule in (1, 2):
            check_garble_eval(ctx, c, KEY, batch, "xf%d" % seed, check_all_wires=(batch <= 130), schedule=schedule, sample=sample)
    except (AssertionError, engine.EngineError) as e:
        bad += 1
        print("FAIL seed", seed, "inputs", ninputs, "gates", ngates, "batch", batch, "schedule", schedule, str(e)[:160])
print("done, failures:", bad)

# ---- second pass: the device-resident pipeline (gc_batch_*: garble -> select -> eval -> decode) on random circuits, random
# key sizes, batches up to 16 500: decoded bits against plaintext evaluation for every instance, tables and R of sampled
# instances against the oracle
from tests.test_gpu_garble_eval import oracle_instance, rnd_for
from tests.util import drbg
from tests.test_gpu_fuzz import device_pipeline_case


def run_device_pass(ctx, n):
    bad = 0
    for seed in range(n):
        try:
            device_pipeline_case(ctx, seed)
        except (AssertionError, engine.EngineError) as e:
            bad += 1
            print("FAIL(device) seed", seed, str(e)[:200])
    print("device pipeline done, failures:", bad)
    return bad


run_device_pass(ctx, N)

# ---- third pass: the streaming garbler / evaluator on random chained programs against the oracle's restatement
import oracle
bad3 = 0
for seed in range(max(1, N // 4)):
    rng = np.random.default_rng(91000 + seed)
    base = int(rng.choice([0, 300, 0xff00, 0x10000, 70000]))
    nsteps = int(rng.integers(1, 9))
    key = drbg("sk%d" % seed, int(rng.choice([16, 24, 32])))
    steps, prim, avail = [], [], []
    nextid = base
    for k in range(nsteps):
        ninputs = int(rng.integers(2, 40))
        if steps and rng.random() < 0.5:  # a circuit of an earlier step under a new binding: the evaluator recognises
            c = steps[int(rng.integers(0, len(steps)))][0]  # the block by its byte skeleton when the ids repeat alike
        else:
            c = random_circuit(rng, ninputs, int(rng.integers(1, 1500)), p_xor=float(rng.choice([0.3, 0.7, 0.9])),
                               reuse=0.0, nout=int(rng.integers(1, 20)))
        in_ = []
        for i in range(c.num_inputs):  # earlier results or fresh primary inputs
            if avail and rng.random() < 0.5:
                in_.append(int(rng.choice(avail)))
            else:
                in_.append(nextid); prim.append(nextid); nextid += 1
        out_ = list(range(nextid, nextid + c.num_outputs)); nextid += c.num_outputs
        # only outputs a gate really writes: an output wire that is an input wire of the circuit is never Set, and a
        # never-set global wire reads as (L0, L1) = (0, 0) in the reference but as (0, R) here (L1 is always L0 ^ R)
        avail += [o for j, o in enumerate(out_) if c.NumWires - c.num_outputs + j >= c.num_inputs]
        steps.append((c, in_, out_))
    try:
        rnd = drbg("sr%d" % seed, 16 * (len(prim) + 1))
        og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
        oe, ge = oracle.StreamEval(key), engine.StreamEval(ctx, key)
        for w in prim:
            l = gg.get(w)["l0"]
            ge.set(w, l); oe.set(w, l)
        for c, in_, out_ in steps:
            want = og.garble(c.Gates, c.NumWires, in_, out_)
            got = gg.garble(c.Gates, c.NumWires, in_, out_)
            assert got == want, "stream bytes"
            nw = max(max(in_), max(out_)) + 1
            assert ge.circuit(c.NumGates, c.NumWires, nw, got) == len(got)
            assert oe.circuit(c.NumGates, c.NumWires, nw, want) == len(want)
            for o in out_:
                assert ge.get(o) == oe.get(o), "evaluated wire %d" % o
        gg.close(); ge.close()
    except (AssertionError, engine.EngineError) as e:
        bad3 += 1
        print("FAIL(stream) seed", seed, "base", base, "steps", nsteps, str(e)[:160])
print("streaming done, failures:", bad3)

# ---- fourth pass: IKNP call sequences (stream positions that are not multiples of 16 bytes, ragged last chunks, the
# device-resident and the host-buffer entry points mixed on one pair) + COT pads against the oracle
from tests.test_gpu_ot import base_setup
bad4 = 0
for seed in range(max(1, N // 8)):
    rng = np.random.default_rng(55000 + seed)
    base, delta, k0 = base_setup("xot%d" % seed)
    orcv, osnd = oracle.IKNPReceiver(base), oracle.IKNPSender(delta, k0)
    grcv, gsnd = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    try:
        for call in range(int(rng.integers(1, 7))):
            n = int(rng.choice([1, 3, 8, 9, 100, 511, 512, 513, 700, 1025, 4000, 20000]))
            b = (np.frombuffer(drbg("xotb%d/%d" % (seed, call), n), np.uint8) & 1).astype(np.uint8)
            wu, wgot = orcv.receive(b)
            wsent = osnd.send(wu, n)
            if rng.random() < 0.5:  # host buffers
                u, got = grcv.receive(b)
                sent = gsnd.send(u, n)
            else:  # device-resident
                chunks = (n + 511) // 512
                packed = np.zeros(chunks * 64, np.uint8)
                pk = np.packbits(b, bitorder="little")
                packed[:len(pk)] = pk
                d_choice = ctx.to_device(packed)
                d_u = ctx.zeros(chunks * 8192)
                d_lr = ctx.zeros((n, 16))
                d_ls = ctx.zeros((n, 16))
                grcv.receive_dev(d_choice, n, d_u, d_lr)
                gsnd.send_dev(d_u, n, d_ls)
                ctx.sync()
                u = d_u.numpy()[: len(wu)].tobytes()
                got = d_lr.numpy().view(np.uint64).reshape(n, 2)
                sent = d_ls.numpy().view(np.uint64).reshape(n, 2)
                got = np.rec.fromarrays([got[:, 0], got[:, 1]], names="d0,d1")
                sent = np.rec.fromarrays([sent[:, 0], sent[:, 1]], names="d0,d1")
            assert bytes(u) == bytes(wu), "u matrix, call %d n %d" % (call, n)
            assert (np.asarray(got["d0"]) == wgot["d0"]).all() and (np.asarray(got["d1"]) == wgot["d1"]).all(), "receiver labels"
            assert (np.asarray(sent["d0"]) == wsent["d0"]).all() and (np.asarray(sent["d1"]) == wsent["d1"]).all(), "sender labels"
    except (AssertionError, engine.EngineError) as e:
        bad4 += 1
        print("FAIL(iknp) seed", seed, str(e)[:160])
    grcv.close(); gsnd.close()
print("iknp done, failures:", bad4)

# ---- fifth pass: circuits whose live labels exceed every LDS plan (HBM-wire kernels: grouped passes, single-pass levels)
# on random level shapes, every gate type, batches that give tiles of 1 to 8 instances; sampled instances byte for byte
from mpc_amd.circuit import synthetic_levelised
bad5 = 0
for seed in range(max(2, N // 12)):
    rng = np.random.default_rng(77000 + seed)
    width = int(rng.choice([48, 64, 300, 1500, 2600, 5000]))
    levels = int(max(2, rng.integers(20000, 45000) // width))
    c = synthetic_levelised(levels, width, float(rng.choice([0.1, 0.3, 0.6])), seed=int(rng.integers(1, 1 << 30)),
                            ninputs=int(rng.choice([16, 64, 200])), or_frac=float(rng.choice([0.0, 0.05])),
                            inv_frac=float(rng.choice([0.0, 0.1])), xnor_frac=float(rng.choice([0.0, 0.1])))
    batch = int(rng.choice([1, 3, 70, 300, 1030, 2100]))
    sample = None if batch <= 3 else sorted(set([0, 1, batch // 2, batch - 2, batch - 1]))
    try:
        check_garble_eval(ctx, c, drbg("hk%d" % seed, int(rng.choice([16, 24, 32]))), batch, "xh%d" % seed,
                          check_all_wires=(batch <= 3), schedule=1, sample=sample)
    except (AssertionError, engine.EngineError) as e:
        bad5 += 1
        print("FAIL(hbm wires) seed", seed, "levels", levels, "width", width, "batch", batch, str(e)[:160])
print("hbm-wire circuits done, failures:", bad5)

# ---- sixth pass: streams of BIG steps (one cooperative launch per step that also gathers / scatters the store labels;
# uploads and serialiser on their own streams, two table buffers in flight): random level shapes and gate mixes, few to
# many thousand input wires (the in-kernel gather then spans workgroups), 16- and 32-bit ids, queued 1 to 4 ahead, repeated
# circuits under new bindings, against the oracle on both sides
bad6 = 0
for seed in range(max(2, N // 10)):
    rng = np.random.default_rng(66000 + seed)
    key = drbg("bk%d" % seed, int(rng.choice([16, 24, 32])))
    shapes = []
    for v in range(int(rng.integers(1, 4))):
        width = int(rng.choice([1100, 2048, 3000, 6000]))
        levels = int(max(3, rng.integers(34000, 70000) // width))
        shapes.append(synthetic_levelised(levels, width, float(rng.choice([0.1, 0.25, 0.5])), seed=int(rng.integers(1, 1 << 30)),
                                          ninputs=int(rng.choice([2, 64, 256, 1000, 9000])), or_frac=float(rng.choice([0.0, 0.03])),
                                          inv_frac=float(rng.choice([0.0, 0.05])), xnor_frac=float(rng.choice([0.0, 0.1]))))
    nextid = int(rng.choice([0, 0xfe00, 0x10000]))
    steps, prim, avail = [], [], []
    for k in range(int(rng.integers(2, 8))):
        c = shapes[int(rng.integers(0, len(shapes)))]
        in_ = []
        for i in range(c.num_inputs):
            if avail and rng.random() < 0.5:
                in_.append(int(avail[int(rng.integers(0, len(avail)))]))
            else:
                in_.append(nextid); prim.append(nextid); nextid += 1
        out_ = list(range(nextid, nextid + c.num_outputs)); nextid += c.num_outputs
        avail += out_
        steps.append((c, in_, out_))
    window = int(rng.integers(1, 5))
    try:
        rnd = drbg("br%d" % seed, 16 * (len(prim) + 1))
        og, gg = oracle.Stream(key, rnd, prim), engine.Stream(ctx, key, rnd, prim)
        oe, ge = oracle.StreamEval(key), engine.StreamEval(ctx, key)
        for w in prim:
            l = gg.get(w)["l0"]
            ge.set(w, l); oe.set(w, l)
        want = [og.garble(c.Gates, c.NumWires, in_, out_) for c, in_, out_ in steps]
        got, issued = [], 0
        for k in range(len(steps)):
            while issued < min(len(steps), k + window):
                c, in_, out_ = steps[issued]
                gg.garble_begin(c.Gates, c.NumWires, in_, out_)
                issued += 1
            got.append(gg.garble_finish())
        assert got == want, "stream bytes"
        for (c, in_, out_), b in zip(steps, got):
            nw = max(max(in_), max(out_)) + 1
            assert ge.circuit(c.NumGates, c.NumWires, nw, b) == len(b)
            assert oe.circuit(c.NumGates, c.NumWires, nw, b) == len(b)
        for o in steps[-1][2][:16] + steps[0][2][:4]:
            assert ge.get(o) == oe.get(o), "evaluated label of wire %d" % o
        ctx.sync()
        gg.close(); ge.close()
    except (AssertionError, engine.EngineError) as e:
        bad6 += 1
        print("FAIL(big steps) seed", seed, "steps", len(steps), "window", window, str(e)[:160])
print("big-step streams done, failures:", bad6)

# ---- seventh pass (round 4): QUEUED programs across the scheduling classes — step groups on the ctx stream, deep steps on
# their lanes (the threshold lowered so that random circuits of a few dozen dependent phases count as deep), now and then a
# big step; results overwrite live wires (write-after-write, write-after-read against steps still in flight on another
# stream), operands repeat, some steps update their inputs in place; 1 to 300 steps queued ahead, 0 to 3 lanes, by handle or
# by content; every byte in program order, every wire afterwards and the evaluator's labels against the oracle
from tests.queued_case import run_case
bad7 = 0
for seed in range(max(2, N // 6)):
    try:
        run_case(seed)
    except (AssertionError, engine.EngineError) as e:
        bad7 += 1
        print("FAIL(queued) seed", seed, str(e)[:300])
print("queued programs across the scheduling classes done, failures:", bad7)
