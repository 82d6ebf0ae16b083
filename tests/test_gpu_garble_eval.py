"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, driven through the C ABI,
must be bit-identical to the CPU oracle / the committed golden digests.

Mirrors the reference's own checks: garble -> eval -> decoded outputs equal the plaintext result
(compiler/arithmetic_test.go:102-151, sha2pc/sha2pc_test.go:15-71), error behaviour of
Circuit.Garble / Circuit.Eval, and the edge cases (batch sizes that are not wave multiples,
all gate types, width-1 chains, wire reuse)."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

import oracle
from mpc_amd import engine
from mpc_amd.circuit import GATE, LABEL, Circuit, and_chain, comparator64, synthetic_levelised
from tests.util import bits_lsb, bits_to_bytes_little, bytes_to_bits_little, drbg, int_from_bits

pytestmark = pytest.mark.gpu

KEY256 = bytes(range(32))
KEY128 = b"0123456789abcdef"  # circuit/garble_bench_test.go:35


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context(0)
    yield c
    c.close()


def rnd_for(c, seed, batch):
    return drbg(seed, 16 * (c.num_inputs + 1) * batch)


def oracle_instance(c, key, rnd, i):
    stride = 16 * (c.num_inputs + 1)
    return oracle.garble(c.Gates, c.NumWires, c.num_inputs, key, rnd[i * stride:(i + 1) * stride])


SCHEDULES = [0, 1, 2]  # 0 = one launch per level, 1 = fused (staggered halves), 2 = fused single-phase (gcengine.h)


def check_garble_eval(ctx, c, key, batch, seed, check_all_wires=True, schedule=1, sample=None):
    """host-buffer API vs oracle, every instance (or the sampled ones), every byte"""
    dc = engine.DeviceCircuit(ctx, c, schedule=schedule)
    rnd = rnd_for(c, seed, batch)
    g = dc.garble(key, rnd, batch=batch, want_wires=check_all_wires, want_io=True)
    if check_all_wires:
        # Garbled.Wires makes the fused schedule walk the XOR levels (every wire is materialised); without it
        # the flattened kernels run.  Same bytes either way.
        g2 = dc.garble(key, rnd, batch=batch, want_wires=False, want_io=True)
        assert (g2["R"] == g["R"]).all() and (g2["slab"] == g["slab"]).all() and (g2["io"] == g["io"]).all()
    bits = (np.frombuffer(drbg(seed + "/bits", c.num_inputs * batch), np.uint8) & 1).reshape(batch, c.num_inputs)
    wires = np.zeros((batch, c.NumWires), LABEL)
    refs = {}
    nin, nout = c.num_inputs, c.num_outputs
    which = range(batch) if sample is None else sample
    for i in which:
        ref = oracle_instance(c, key, rnd, i)
        refs[i] = ref
        assert g["R"][i] == ref["R"], "R of instance %d" % i
        assert (g["slab"][i] == ref["slab"]).all(), "slab of instance %d" % i
        if check_all_wires:
            assert (g["wires"][i] == ref["wires"]).all(), "wires of instance %d" % i
        assert (g["io"][i][:nin] == ref["wires"][:nin]).all()
        assert (g["io"][i][nin:] == ref["wires"][c.NumWires - nout:]).all()
        wires[i, :nin] = np.where(bits[i].astype(bool), ref["wires"]["l1"][:nin], ref["wires"]["l0"][:nin])
    if sample is not None:  # unsampled instances: any valid input labels (the garbler's zero labels)
        rest = np.setdiff1d(np.arange(batch), np.asarray(list(sample)))
        wires[rest, :nin] = g["io"][rest, :nin]["l0"]
    inputs = wires[:, :nin].copy()
    out = dc.eval(key, g["slab"], wires=wires, batch=batch)
    for i in which:
        w = np.zeros(c.NumWires, LABEL)
        w[:nin] = inputs[i]
        oracle.eval_(c.Gates, c.NumWires, key, w, refs[i]["slab"])
        assert (wires[i] == w).all(), "evaluated wires of instance %d" % i
        assert (out[i] == w[c.NumWires - nout:]).all()
        plain = oracle.compute(c.Gates, c.NumWires, nin, bits[i])
        for j in range(nout):  # BitFromLabel
            wi = c.NumWires - nout + j
            want = refs[i]["wires"][wi]["l1"] if plain[wi] else refs[i]["wires"][wi]["l0"]
            assert out[i][j] == want
    # inputs-only entry (the evaluator normally holds just its input labels)
    out2 = dc.eval(key, g["slab"], inputs=inputs, batch=batch)
    assert (out2 == out).all()
    dc.close()


@pytest.mark.parametrize("schedule", SCHEDULES)
@pytest.mark.parametrize("batch", [1, 2, 3, 5, 33, 64, 65, 100, 255, 256, 257, 300, 513, 1030])
def test_all_gate_types_ragged_batches(ctx, batch, schedule):
    c = synthetic_levelised(10, 48, 0.3, seed=5, ninputs=40, or_frac=0.1, inv_frac=0.1, xnor_frac=0.1)
    check_garble_eval(ctx, c, KEY256, batch, "ragged%d" % batch, schedule=schedule)


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_wide_levels(ctx, schedule):
    # levels wider than one 1024-thread workgroup pass, every tile size of the fused schedule
    c = synthetic_levelised(4, 700, 0.4, seed=13, ninputs=64, or_frac=0.05, inv_frac=0.1, xnor_frac=0.05)
    for batch in (3, 520, 2100):
        check_garble_eval(ctx, c, KEY128, batch, "wide%d" % batch, check_all_wires=(batch < 1000), schedule=schedule)


@pytest.mark.parametrize("schedule", [1, 2])
@pytest.mark.parametrize("batch", [4099, 16389])
def test_large_tiles(ctx, add64_circ, batch, schedule):
    # tiles of 8 and 64 instances per workgroup (small live set): halves of 4 / 32 instances
    sample = list(range(0, batch, 37)) + list(range(batch - 70, batch))
    check_garble_eval(ctx, add64_circ, KEY128, batch, "tiles%d" % batch, schedule=schedule, sample=sorted(set(sample)))


@pytest.mark.parametrize("schedule", SCHEDULES)
@pytest.mark.parametrize("key", [KEY128, bytes(range(7, 31)), KEY256], ids=["aes128", "aes192", "aes256"])
def test_key_sizes(ctx, key, schedule):
    c = synthetic_levelised(6, 70, 0.4, seed=9, ninputs=32, or_frac=0.1, inv_frac=0.1, xnor_frac=0.1)
    check_garble_eval(ctx, c, key, 70, "keys%d" % len(key), schedule=schedule)


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_and_chain_width_one(ctx, schedule):
    # buildANDChain (garble_bench_test.go:19-33): depth n, width 1 — one launch per gate
    check_garble_eval(ctx, and_chain(300), KEY128, 17, "chain", schedule=schedule)


def test_comparator64_millionaire(ctx):
    c = comparator64()
    for schedule in SCHEDULES:
        check_garble_eval(ctx, c, KEY256, 9, "cmp", schedule=schedule)
    dc = engine.DeviceCircuit(ctx, c)
    for a, b in ((750000, 800000), (900000, 800000)):  # README.md:57-108
        rnd = rnd_for(c, "mill", 1)
        g = dc.garble(KEY256, rnd, batch=1)
        bits = np.concatenate([bits_lsb(a, 64), bits_lsb(b, 64)]).astype(bool)
        inp = np.where(bits, g["io"][0]["l1"][:128], g["io"][0]["l0"][:128])
        out = dc.eval(KEY256, g["slab"], inputs=inp[None, :], batch=1)
        ow = g["io"][0][128]
        assert out[0][0] == (ow["l1"] if a > b else ow["l0"])
    dc.close()


def test_wire_reuse(ctx):
    g = np.zeros(3, GATE)
    g[0] = (0, 1, 2, 0, 0)
    g[1] = (2, 0, 2, 2, 0)
    g[2] = (2, 1, 3, 0, 0)
    for schedule in SCHEDULES:
        check_garble_eval(ctx, Circuit(4, [1, 1], [1], g), KEY256, 7, "reuse", schedule=schedule)


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_add64(ctx, add64_circ, schedule):
    check_garble_eval(ctx, add64_circ, KEY256, 40, "add64", schedule=schedule)


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_aes128_circuit_batch(ctx, aes_circ, schedule):
    check_garble_eval(ctx, aes_circ, KEY256, 6, "aes", check_all_wires=True, schedule=schedule)


def test_sha256xor_reference_digest(ctx, sha_circ):
    # sha2pc/sha2pc_test.go:124: a[i]=i, b[i]=32-i -> 4b2f7457...; instance 1 random
    c = sha_circ
    dc = engine.DeviceCircuit(ctx, c)
    assert dc.info.slab_rows == 42914  # sha2pc/params.go:26
    batch = 2
    key = drbg("shakey", 32)
    g = dc.garble(key, rnd_for(c, "sha", batch), batch=batch)
    a = [bytes(range(32)), drbg("a1", 32)]
    b = [bytes(32 - i for i in range(32)), drbg("b1", 32)]
    inputs = np.zeros((batch, 512), LABEL)
    for i in range(batch):
        bits = np.concatenate([bytes_to_bits_little(a[i]), bytes_to_bits_little(b[i])]).astype(bool)
        inputs[i] = np.where(bits, g["io"][i]["l1"][:512], g["io"][i]["l0"][:512])
    out = dc.eval(key, g["slab"], inputs=inputs, batch=batch)
    for i in range(batch):
        ow = g["io"][i][512:]
        bits = []
        for j in range(256):
            assert out[i][j] == ow[j]["l0"] or out[i][j] == ow[j]["l1"]
            bits.append(1 if out[i][j] == ow[j]["l1"] else 0)
        want = hashlib.sha256(bytes(x ^ y for x, y in zip(a[i], b[i]))).hexdigest()
        assert bits_to_bytes_little(bits).hex() == want
        if i == 0:
            assert want == "4b2f74579fc7c778745121996f604371a326dc5174f9851706032626668abf2e"
    dc.close()


def test_golden_digests(ctx, golden_dir):
    """committed fixtures (tests/golden/garble_golden.json, made by make_golden.py with the oracle)"""
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = json.load(open(os.path.join(golden_dir, "garble_golden.json")))
    for name, c in mk.circuits().items():
        for schedule in SCHEDULES:
          dc = engine.DeviceCircuit(ctx, c, schedule=schedule)
          n = mk.COUNTS[name]
          for kn, kh in gold["keys"].items():
              key = bytes.fromhex(kh)
              streams = [mk.instance_streams(name, c, i) for i in range(n)]
              rnd = b"".join(s[0] for s in streams)
              g = dc.garble(key, rnd, batch=n)
              nin, nout = c.num_inputs, c.num_outputs
              wires = np.zeros((n, c.NumWires), LABEL)
              for i in range(n):
                  h = hashlib.sha256()
                  h.update(g["R"][i:i + 1].tobytes())
                  h.update(np.ascontiguousarray(g["slab"][i]).tobytes())
                  h.update(np.ascontiguousarray(g["io"][i]["l0"][nin:]).tobytes())
                  assert h.hexdigest() == gold["circuits"][name][kn][i]["garble"], (name, kn, i)
                  wires[i, :nin] = np.where(streams[i][1].astype(bool), g["io"][i]["l1"][:nin], g["io"][i]["l0"][:nin])
              dc.eval(key, g["slab"], wires=wires, batch=n)
              for i in range(n):
                  assert hashlib.sha256(wires[i].tobytes()).hexdigest() == gold["circuits"][name][kn][i]["eval"]
          dc.close()


def test_error_behaviour(ctx, add64_circ):
    c = add64_circ
    dc = engine.DeviceCircuit(ctx, c)
    rnd = rnd_for(c, "err", 1)
    for klen in (0, 15, 17, 33):  # aes.NewCipher (garble.go:260)
        with pytest.raises(engine.EngineError) as e:
            dc.garble(bytes(klen), rnd)
        assert e.value.code == engine.GC_E_KEYSIZE
        assert "invalid key size" in str(e.value)
    with pytest.raises(engine.EngineError) as e:  # io.Reader runs dry (garble.go:272)
        dc.garble(KEY256, rnd[:-1])
    assert e.value.code == engine.GC_E_RAND
    with pytest.raises(engine.EngineError) as e:  # R is read before the cipher is built (garble.go:253-260)
        dc.garble(bytes(5), rnd[:8])
    assert e.value.code == engine.GC_E_RAND
    g = dc.garble(KEY256, rnd)
    inp = g["io"][0]["l0"][: c.num_inputs][None, :]
    with pytest.raises(engine.EngineError) as e:  # corrupted circuit: rows missing (eval.go:54-56)
        dc.eval(KEY256, np.ascontiguousarray(g["slab"][:, :-1]), inputs=inp, batch=1)
    assert e.value.code == engine.GC_E_ROWS
    with pytest.raises(engine.EngineError) as e:
        dc.eval(bytes(3), g["slab"], inputs=inp, batch=1)
    assert e.value.code == engine.GC_E_KEYSIZE
    dc.close()
    bad = np.zeros(1, GATE)
    bad[0] = (0, 1, 2, 6, 0)
    with pytest.raises(engine.EngineError) as e:  # "invalid gate type" (garble.go:326)
        engine.DeviceCircuit(ctx, Circuit(3, [1, 1], [1], bad))
    assert e.value.code == engine.GC_E_GATE


def test_graph_and_direct_launch_agree(ctx):
    c = synthetic_levelised(12, 64, 0.25, seed=21, ninputs=64, inv_frac=0.05)
    dc = engine.DeviceCircuit(ctx, c)
    batch = 128
    rnd = rnd_for(c, "graph", batch)
    d_rnd = ctx.to_device(rnd)
    slabs = []
    for graph in (True, False, True):
        b = engine.Batch(dc, batch)
        b.set_schedule(0)
        b.set_graph(graph)
        b.garble(KEY256, d_rnd)
        b.garble(KEY256, d_rnd)  # second call replays the captured graph
        slabs.append(b.read_slab().copy())
        assert b.last_ms > 0 and b.last_launches == dc.info.n_steps  # gate kernels only
        b.close()
    assert (slabs[0] == slabs[1]).all() and (slabs[0] == slabs[2]).all()
    ref = oracle_instance(c, KEY256, rnd, 77)
    assert (slabs[0][77] == ref["slab"]).all()
    dc.close()


@pytest.mark.parametrize("schedule", SCHEDULES)
def test_device_resident_pipeline_full_size(ctx, aes_circ, schedule):
    """BASELINE config 2 shape: AES-128 circuit x 1024 instances, device-resident API.
    Size-independent property: decoded outputs == AES-128(key, pt) for every instance; plus byte
    parity with the oracle on sampled instances."""
    c = aes_circ
    batch = 1024
    dc = engine.DeviceCircuit(ctx, c)
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    gb.set_schedule(schedule)
    ev.set_schedule(schedule)
    rnd = rnd_for(c, "full", batch)
    d_rnd = ctx.to_device(rnd)
    keys = [drbg("k%d" % i, 16) for i in range(batch)]
    pts = [drbg("p%d" % i, 16) for i in range(batch)]
    bits = np.zeros((batch, 256), np.uint8)
    for i in range(batch):
        bits[i, :128] = bits_lsb(int.from_bytes(keys[i], "big"), 128)
        bits[i, 128:] = bits_lsb(int.from_bytes(pts[i], "big"), 128)
    d_bits = ctx.to_device(bits)
    d_out = ctx.zeros((batch, 128))
    d_mis = ctx.zeros(1, np.int32)
    gb.garble(KEY256, d_rnd)
    ev.select_inputs(gb, d_bits)
    ev.eval(KEY256, gb)
    gb.decode(ev, d_out, d_mis)
    ctx.sync()
    assert int(d_mis.numpy()[0]) == 0
    out = d_out.numpy()
    for i in range(batch):
        ct = int_from_bits(out[i]).to_bytes(16, "big")
        assert ct == oracle.aes_encrypt(keys[i], pts[i]), "instance %d" % i
    slab = gb.read_slab()
    R = gb.read_r()
    for i in (0, 1, 63, 64, 511, 1023):
        ref = oracle_instance(c, KEY256, rnd, i)
        assert R[i] == ref["R"] and (slab[i] == ref["slab"]).all()
    assert gb.last_ms > 0 and ev.last_ms > 0
    gb.close(); ev.close(); dc.close()


def test_pipeline_graph_replay(ctx, add64_circ):
    """gc_ctx_capture_*: the garble -> select -> eval -> decode sequence recorded once and replayed gives the bytes of
    the direct calls, also after the inputs (same device buffers) changed."""
    c = add64_circ
    batch = 600
    dc = engine.DeviceCircuit(ctx, c)
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    d_rnd = ctx.to_device(rnd_for(c, "graph0", batch))
    d_bits = ctx.zeros((batch, c.num_inputs))
    d_out = ctx.zeros((batch, c.num_outputs))
    d_mis = ctx.zeros(1, np.int32)

    def step():
        gb.garble(KEY256, d_rnd)
        ev.select_inputs(gb, d_bits)
        ev.eval(KEY256, gb)
        gb.decode(ev, d_out, d_mis)

    step()  # uploads the key; not capturable
    ctx.sync()
    g = ctx.capture(step)
    for rep in range(2):
        rnd = rnd_for(c, "graph%d" % (rep + 1), batch)
        bits = (np.frombuffer(drbg("gbits%d" % rep, batch * c.num_inputs), np.uint8) & 1).reshape(batch, -1)
        d_rnd.upload(np.frombuffer(rnd, np.uint8))
        d_bits.upload(bits)
        g.launch()
        ctx.sync()
        assert int(d_mis.numpy()[0]) == 0
        out = d_out.numpy()
        slab, R = gb.read_slab(), gb.read_r()
        for i in (0, 1, 299, 599):
            ref = oracle_instance(c, KEY256, rnd, i)
            assert R[i] == ref["R"] and (slab[i] == ref["slab"]).all()
            plain = oracle.compute(c.Gates, c.NumWires, c.num_inputs, bits[i])
            assert (out[i] == plain[c.NumWires - c.num_outputs:]).all()
    g.close(); gb.close(); ev.close(); dc.close()


def test_concurrent_host_calls_on_one_circuit(ctx, add64_circ):
    """Circuit.Garble may be called from several goroutines on one *Circuit (atomic scratch pool, garble.go:195-225);
    here: eight threads on one gc_circ (pooled batches, calls serialised on the ctx's stream) and on a second ctx"""
    import threading
    c = add64_circ
    dc = engine.DeviceCircuit(ctx, c)
    ctx2 = engine.Context(0)
    dc2 = engine.DeviceCircuit(ctx2, c)
    errors = []

    def worker(t):
        try:
            d = dc2 if t % 4 == 3 else dc
            for it in range(6):
                batch = [1, 3, 70, 300][(t + it) % 4]
                rnd = rnd_for(c, "thr%d/%d" % (t, it), batch)
                g = d.garble(KEY128, rnd, batch=batch, want_wires=(it % 2 == 0), want_io=True)
                for i in (0, batch - 1):
                    ref = oracle_instance(c, KEY128, rnd, i)
                    assert g["R"][i] == ref["R"] and (g["slab"][i] == ref["slab"]).all()
                    if it % 2 == 0:
                        assert (g["wires"][i] == ref["wires"]).all()
                inputs = np.ascontiguousarray(g["io"][:, : c.num_inputs]["l0"])
                out = d.eval(KEY128, g["slab"], inputs=inputs, batch=batch)
                plain = oracle.compute(c.Gates, c.NumWires, c.num_inputs, np.zeros(c.num_inputs, np.uint8))
                for j in range(c.num_outputs):
                    wi = c.NumWires - c.num_outputs + j
                    want = g["io"][0][c.num_inputs + j]["l1"] if plain[wi] else g["io"][0][c.num_inputs + j]["l0"]
                    assert out[0][j] == want
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append("thread %d: %r" % (t, e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dc.close(); dc2.close(); ctx2.close()
    assert not errors, errors[:3]


def test_host_api_pinned_buffers_pipeline(ctx, aes_circ):
    """gc_garble / gc_eval on PINNED caller memory (gc_host_alloc): direct DMA, chunk-pipelined against the layout
    transposes (several chunks at this size) — same bytes as the pageable path and as the oracle"""
    import ctypes as C
    c = aes_circ
    batch = 300
    dc = engine.DeviceCircuit(ctx, c)
    L = engine.lib()
    rows, nin, nout = dc.info.slab_rows, c.num_inputs, c.num_outputs
    rnd = np.frombuffer(rnd_for(c, "pinned", batch), np.uint8).copy()
    k = np.frombuffer(KEY256, np.uint8).copy()
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    keep = [engine.PinnedArray((batch,), LABEL), engine.PinnedArray((batch, rows), LABEL),
            engine.PinnedArray((batch, nin + nout), engine.WIRE), engine.PinnedArray((batch, c.NumWires), LABEL),
            engine.PinnedArray((batch, nout), LABEL)]
    R, slab, io, wires, outl = [x.a for x in keep]
    assert L.gc_host_is_pinned(ptr(slab)) == 1 and L.gc_host_is_pinned(ptr(rnd)) == 0
    assert L.gc_garble(dc.h, ptr(k), len(k), ptr(rnd), len(rnd), batch, ptr(R), None, ptr(io), ptr(slab)) == 0
    ref = dc.garble(KEY256, rnd.tobytes(), batch=batch, want_wires=False, want_io=True)  # pageable
    assert (R == ref["R"]).all() and (slab == ref["slab"]).all() and (io == ref["io"]).all()
    for i in (0, 159, 160, 299):  # chunk boundaries
        o = oracle_instance(c, KEY256, rnd.tobytes(), i)
        assert R[i] == o["R"] and (slab[i] == o["slab"]).all()
    # eval: pinned slab + pinned full wire array (strided 2-D input copy, every wire read back)
    wires[...] = np.zeros((), LABEL)
    wires[:, :nin] = io[:, :nin]["l0"]
    assert L.gc_eval(dc.h, ptr(k), len(k), batch, ptr(wires), None, ptr(slab), rows, ptr(outl)) == 0
    out_ref = dc.eval(KEY256, ref["slab"], inputs=np.ascontiguousarray(ref["io"][:, :nin]["l0"]), batch=batch)
    assert (outl == out_ref).all() and (wires[:, c.NumWires - nout:] == out_ref).all()
    for i in (0, 160, 299):
        w = np.zeros(c.NumWires, LABEL)
        w[:nin] = io[i, :nin]["l0"]
        oracle.eval_(c.Gates, c.NumWires, KEY256, w, slab[i])
        assert (wires[i] == w).all()
    # a registered (not allocated) range works the same
    own = np.zeros((batch, rows), LABEL)
    assert L.gc_host_register(ptr(own), own.nbytes) == 0 and L.gc_host_is_pinned(ptr(own)) == 1
    assert L.gc_garble(dc.h, ptr(k), len(k), ptr(rnd), len(rnd), batch, ptr(R), None, None, ptr(own)) == 0
    assert (own == slab).all()
    assert L.gc_host_unregister(ptr(own)) == 0
    dc.close()
    for x in keep:
        x.close()


@pytest.mark.parametrize("batch", [1, 3, 70, 1030])
def test_global_wire_kernels_on_a_circuit_that_exceeds_lds(ctx, batch):
    """a circuit whose live labels do not fit the LDS plans runs the fused kernels with wires in HBM (passes of a level
    grouped: descriptors, operands, hashes); every gate type, levels of several passes, all wires compared.  batch = 1
    takes the gate-parallel form: one launch per level, pass k of the level = workgroup k, replayed as a hipGraph."""
    c = synthetic_levelised(14, 2500, 0.3, seed=77, ninputs=64, or_frac=0.05, inv_frac=0.08, xnor_frac=0.1)
    dc = engine.DeviceCircuit(ctx, c)
    b = engine.Batch(dc, batch)
    assert not b.lds_wires, "expected the HBM-wire kernels (live labels %d)" % dc.info.n_flat_slots
    b.close(); dc.close()
    sample = None if batch < 10 else sorted(set(list(range(0, batch, 97)) + [batch - 1, batch - 2]))
    check_garble_eval(ctx, c, KEY256, batch, "glob%d" % batch, check_all_wires=(batch <= 70), schedule=1, sample=sample)
    if batch == 1:  # again: the second call replays the recorded graph; another key size records its own
        check_garble_eval(ctx, c, KEY256, batch, "glob1b", check_all_wires=True, schedule=1)
        check_garble_eval(ctx, c, KEY128, batch, "glob1c", check_all_wires=False, schedule=1)


def test_global_wire_kernels_hash_and_free_waves(ctx):
    """wide levels split the workgroup into hash waves and free waves (fused_kernels.hip): besides mixed levels, levels
    without a hashed gate (all sixteen waves stream), without a free gate (all sixteen hash), with a handful of ANDs (fewer
    hash lanes than one wave) and with a handful of XORs; tile sizes 1 and 4; every wire compared"""
    from mpc_amd.circuit import AND, XNOR, XOR, Circuit
    base = synthetic_levelised(12, 2500, 0.3, seed=177, ninputs=64, or_frac=0.05, inv_frac=0.08, xnor_frac=0.1)
    g = base.Gates.copy()
    w = 2500
    g["op"][2 * w:3 * w] = np.where(np.arange(w) % 7 == 0, XNOR, XOR)   # no hashed gate
    g["op"][4 * w:5 * w] = AND                                          # no free gate
    g["op"][6 * w:7 * w] = XOR
    g["op"][6 * w + 11:6 * w + 16] = AND                                # five ANDs
    g["op"][8 * w:9 * w] = AND
    g["op"][8 * w + 100:8 * w + 105] = XOR                              # five XORs
    c = Circuit(base.NumWires, base.Inputs, base.Outputs, g, "roles")
    dc = engine.DeviceCircuit(ctx, c)
    b = engine.Batch(dc, 5)
    assert not b.lds_wires, "expected the HBM-wire kernels (live labels %d)" % dc.info.n_flat_slots
    b.close(); dc.close()
    check_garble_eval(ctx, c, KEY256, 5, "roles5", check_all_wires=True, schedule=1)
    check_garble_eval(ctx, c, KEY128, 1027, "roles1027", check_all_wires=False, schedule=1, sample=[0, 3, 512, 1023, 1026])


@pytest.mark.parametrize("batch", [3, 520, 1030])
def test_global_wire_kernels_column_sliced_narrow_levels(ctx, batch):
    """a deep circuit of narrow levels WITHOUT OR gates, live labels beyond the LDS plans: every level runs column-sliced
    (four lanes per AES block, k_garble_col / k_eval_col) — ANDs, INVs, XORs and XNORs; at the largest tile size some levels
    need a second pass; every wire against the oracle for the small batch, sampled instances for the others; and the same
    circuit with GC_NO_COL (the wide form) for the small batch"""
    c = synthetic_levelised(700, 64, 0.2, seed=79, ninputs=64, inv_frac=0.08, xnor_frac=0.1)
    assert c.stats()["OR"] == 0
    dc = engine.DeviceCircuit(ctx, c)
    b = engine.Batch(dc, batch)
    assert not b.lds_wires, "expected the HBM-wire kernels (live labels %d)" % dc.info.n_flat_slots
    b.close(); dc.close()
    sample = None if batch < 10 else [0, 1, 255, 256, batch - 2, batch - 1]
    check_garble_eval(ctx, c, KEY256, batch, "col%d" % batch, check_all_wires=(batch < 10), schedule=1, sample=sample)
    if batch < 10:
        check_garble_eval(ctx, c, KEY128, batch, "colk128", check_all_wires=True, schedule=1)
        os.environ["GC_NO_COL"] = "1"
        try:
            check_garble_eval(ctx, c, KEY256, batch, "colwide", check_all_wires=True, schedule=1)
        finally:
            del os.environ["GC_NO_COL"]


@pytest.mark.parametrize("batch", [2, 520])
def test_global_wire_kernels_narrow_deep_circuit(ctx, batch):
    """the same kernels on levels of a single pass (their single-pass instantiation with the descriptor prefetched
    across the level barrier): 900 levels x 64 gates, live labels beyond the LDS plans"""
    c = synthetic_levelised(900, 64, 0.3, seed=78, ninputs=64, or_frac=0.05, inv_frac=0.08, xnor_frac=0.1)
    dc = engine.DeviceCircuit(ctx, c)
    b = engine.Batch(dc, batch)
    assert not b.lds_wires, "expected the HBM-wire kernels (live labels %d)" % dc.info.n_flat_slots
    b.close(); dc.close()
    sample = None if batch < 10 else [0, 1, 255, 256, 519]
    check_garble_eval(ctx, c, KEY128, batch, "globn%d" % batch, check_all_wires=(batch < 10), schedule=1, sample=sample)


@pytest.mark.gpu
def test_late_schedule_circuits_match_oracle(ctx):
    """circuits that only get an LDS plan with the late schedule (plan.cpp: build_flat late; a 112-bit array multiplier keeps
    6 500 labels alive when every gate runs as early as it can): the flattened kernels on that schedule, one instance and
    tiles of several, all three key sizes — tables, wire labels of inputs and outputs and the evaluated labels equal the
    oracle's serial loop (tweaks and table rows follow the gate list, not the schedule)"""
    from mpc_amd.circuit import multiplier
    c = multiplier(112)
    dc = engine.DeviceCircuit(ctx, c)
    assert dc.info.n_flat_slots < 2000 < dc.info.n_lds_slots
    for batch, keylen in ((1, 32), (5, 16), (300, 24), (1030, 32)):
        sample = None if batch <= 5 else [0, 1, batch // 2, batch - 1]
        check_garble_eval(ctx, c, drbg("late%d" % batch, keylen), batch, "late/%d" % batch, check_all_wires=False, sample=sample)
