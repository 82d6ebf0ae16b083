"""GPU parity tests of the IKNP OT-extension kernels and MITCCRH (through the C ABI) against the CPU
oracle, plus the reference's own properties (ot/iknp_test.go:98-113 correlation, ot/ot_test.go:83-97
delivery, ot/mitccrh_test.go:23-30 vectors)."""
import numpy as np
import pytest

import oracle
from mpc_amd import engine
from mpc_amd.circuit import LABEL, WIRE
from tests.test_oracle_kat import MITCCRH_BLOCKS
from tests.util import drbg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context(0)
    yield c
    c.close()


def labels(seed, n):
    raw = drbg(seed, 16 * n)
    out = np.zeros(n, LABEL)
    for i in range(n):
        out[i] = oracle.label_from_bytes(raw[16 * i:16 * i + 16])
    return out


def base_setup(seed):
    base = np.zeros(128, WIRE)
    base["l0"] = labels(seed + "l0", 128)
    base["l1"] = labels(seed + "l1", 128)
    delta = oracle.label_from_bytes(drbg(seed + "delta", 16))
    k0 = np.zeros(128, LABEL)
    for i in range(128):
        k0[i] = base[i]["l1"] if oracle.label_bit(delta, i) else base[i]["l0"]
    return base, delta, k0


# iknp_test.go:32-37 chunk-size cases + ragged / sub-byte / multi-chunk sizes
SIZES = [0, 1, 2, 7, 8, 9, 129, 511, 512, 513, 700, 1024, 1025, 2049, 2560, 65536]


@pytest.mark.parametrize("n", SIZES)
def test_iknp_matches_oracle_and_correlates(ctx, n):
    base, delta, k0 = base_setup("iknp%d" % n)
    b = (np.frombuffer(drbg("b%d" % n, max(n, 1)), np.uint8)[:n] & 1).astype(np.uint8)
    rcv, snd = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    orcv, osnd = oracle.IKNPReceiver(base), oracle.IKNPSender(delta, k0)
    u, got = rcv.receive(b)
    ou, ogot = orcv.receive(b)
    assert u == ou, "u-matrix bytes differ from the oracle"
    assert (got == ogot).all(), "receiver labels differ from the oracle"
    sent = snd.send(u, n)
    assert (sent == osnd.send(ou, n)).all(), "sender labels differ from the oracle"
    if n:
        x0 = sent["d0"] ^ np.where(b == 1, np.uint64(delta[0]), np.uint64(0))
        x1 = sent["d1"] ^ np.where(b == 1, np.uint64(delta[1]), np.uint64(0))
        assert (got["d0"] == x0).all() and (got["d1"] == x1).all()  # rcvd = sent ^ b*delta
    rcv.close(); snd.close()


def test_iknp_streams_persist_across_calls(ctx):
    # column CTR streams continue across calls, also from a mid-block byte offset (iknp.go:488,632-637)
    base, delta, k0 = base_setup("persist")
    rcv, snd = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    orcv, osnd = oracle.IKNPReceiver(base), oracle.IKNPSender(delta, k0)
    for n in (24 * 8, 5, 1000, 513, 77):
        b = (np.frombuffer(drbg("pb%d" % n, n), np.uint8) & 1).astype(np.uint8)
        u, got = rcv.receive(b)
        ou, ogot = orcv.receive(b)
        assert u == ou and (got == ogot).all()
        assert (snd.send(u, n) == osnd.send(ou, n)).all()
    with pytest.raises(engine.EngineError):  # "invalid chunk size" (iknp.go:207-209)
        snd.send(b"\x00" * 100, 64)
    rcv.close(); snd.close()


def test_mitccrh_reference_vectors(ctx):
    blks = np.zeros(16, LABEL)
    out = engine.mitccrh_hash(ctx, (0, 0), 0, blks, 2)  # seed 0, keys 0..7, h = 2, zero blocks
    for i in range(8):
        for j in range(2):
            assert oracle.label_to_bytes(out[2 * i + j]).hex() == MITCCRH_BLOCKS[i]


@pytest.mark.parametrize("n,h", [(1, 1), (8, 2), (13, 1), (1000, 2), (4097, 1)])
def test_mitccrh_matches_oracle(ctx, n, h):
    seed = oracle.label_from_bytes(drbg("ms%d" % n, 16))
    blks = labels("mb%d" % n, n * h)
    got = engine.mitccrh_hash(ctx, seed, 0, blks, h)
    m = oracle.MITCCRH(seed, 8)
    want = blks.copy()
    for i in range(0, n, 8):  # the Go object hashes 8 keys per call (cot.go:160-171)
        k = min(8, n - i)
        pad = np.zeros(8 * h, LABEL)
        pad[: k * h] = want[i * h:(i + k) * h]
        m.hash(pad, 8, h)
        want[i * h:(i + k) * h] = pad[: k * h]
    assert (got == want).all()


@pytest.mark.parametrize("n", [1, 8, 13, 64, 3000])
def test_cot_pipeline_delivers_chosen_label(ctx, n):
    # ot_test.go:83-97 through IKNP + MITCCRH on the GPU; wire bytes equal the oracle's
    base, delta, k0 = base_setup("cot%d" % n)
    flags = np.array([i % 2 for i in range(n)], np.uint8)
    wires = np.zeros(n, WIRE)
    wires["l0"] = labels("w0", n)
    wires["l1"] = labels("w1", n)
    rcv, snd = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    u, got = rcv.receive(flags)
    data = snd.send(u, n)
    seed = oracle.label_from_bytes(drbg("cotseed", 16))
    sent = engine.cot_send_pads(ctx, seed, delta, data, wires)
    assert (sent == oracle.cot_send_pads(seed, delta, data, wires)).all()
    res = engine.cot_receive_unpad(ctx, seed, flags, sent, got)
    assert (res == oracle.cot_receive_unpad(seed, flags, sent, got)).all()
    for i in range(n):
        assert res[i] == (wires[i]["l1"] if flags[i] else wires[i]["l0"])
    rcv.close(); snd.close()


@pytest.mark.parametrize("n", [0, 1, 129, 700, 5000])
def test_kos_check_matches_oracle(ctx, n):
    base, delta, k0 = base_setup("kos%d" % n)
    rcv, snd = oracle.IKNPReceiver(base), oracle.IKNPSender(delta, k0)
    b = (np.frombuffer(drbg("kb%d" % n, max(n, 1)), np.uint8)[:n] & 1).astype(np.uint8)
    u, got = rcv.receive(b)
    sent = snd.send(u, n)
    bcv = (np.frombuffer(drbg("kbcv", 256), np.uint8) & 1).astype(np.uint8)
    u2, cvr = rcv.receive(bcv)
    cvs = snd.send(u2, 256)
    seed2 = oracle.label_from_bytes(drbg("seed2", 16))
    want = oracle.kos_receiver_tags(seed2, got, b, cvr, bcv)
    have = engine.kos_receiver_tags(ctx, seed2, got, b, cvr, bcv)
    assert have == want
    x, t0, t1 = have
    assert engine.kos_sender_check(ctx, seed2, sent, cvs, delta, x, t0, t1)
    assert not engine.kos_sender_check(ctx, seed2, sent, cvs, delta, x, (t0[0] ^ 1, t0[1]), t1)
    if n:
        bad = sent.copy()
        bad[n // 2]["d1"] ^= 1 << 40
        assert not engine.kos_sender_check(ctx, seed2, bad, cvs, delta, x, t0, t1)
    # the device-resident forms: labels and choice bytes stay in HBM (what gc_iknp_*_dev leave there)
    d_got = ctx.to_device(np.ascontiguousarray(got).view(np.uint8).reshape(-1).copy() if n else np.zeros(16, np.uint8))
    d_sent = ctx.to_device(np.ascontiguousarray(sent).view(np.uint8).reshape(-1).copy() if n else np.zeros(16, np.uint8))
    d_b = ctx.to_device(b.copy() if n else np.zeros(1, np.uint8))
    assert engine.kos_receiver_tags_dev(ctx, seed2, d_got, d_b, n, cvr, bcv) == want
    assert engine.kos_sender_check_dev(ctx, seed2, d_sent, n, cvs, delta, x, t0, t1)
    assert not engine.kos_sender_check_dev(ctx, seed2, d_sent, n, cvs, delta, x, t0, (t1[0], t1[1] ^ 2))


@pytest.mark.parametrize("n", [64, 1024, 1000, 70, 513])
def test_bitcot_matches_oracle(ctx, n):
    base, delta, k0 = base_setup("bit%d" % n)
    choices = np.frombuffer(drbg("bc%d" % n, 8 * ((n + 63) // 64)), "<u8").copy()
    rcv, snd = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    orcv, osnd = oracle.IKNPReceiver(base), oracle.IKNPSender(delta, k0)
    u, r = rcv.receive_bits(choices, n)
    ou, orr = oracle.iknp_receive_bits(orcv, choices, n)
    assert u == ou and (r == orr).all()
    s = snd.send_bits(u, n)
    assert (s == oracle.iknp_send_bits(osnd, ou, n)).all()
    if n % 64 == 0:  # bitcot_test.go: s ^ r == Delta.Bit(0) & c
        d0 = np.uint64(0xFFFFFFFFFFFFFFFF) if oracle.label_bit(delta, 0) else np.uint64(0)
        assert ((s ^ r) == (choices & d0)).all()
    rcv.close(); snd.close()


def test_iknp_device_resident_api_matches_host_api(ctx):
    """gc_iknp_receive_dev / gc_iknp_send_dev (HBM in, HBM out, asynchronous) produce the bytes of the host calls,
    including the persistent stream position across calls of ragged sizes."""
    base, delta, k0 = base_setup("dev")
    rx_h, tx_h = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    rx_d, tx_d = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    for n in (700, 512, 5, 3000):
        b = np.frombuffer(drbg("devb%d" % n, n), np.uint8) & 1
        u, lr = rx_h.receive(b)
        ls = tx_h.send(u, n)
        chunks = (n + 511) // 512
        packed = np.zeros(chunks * 64, np.uint8)
        pb = np.packbits(b, bitorder="little")
        packed[:len(pb)] = pb
        d_c = ctx.to_device(packed)
        d_u = ctx.zeros(chunks * 8192)
        d_lr = ctx.zeros((n, 16))
        d_ls = ctx.zeros((n, 16))
        rx_d.receive_dev(d_c, n, d_u, d_lr)
        tx_d.send_dev(d_u, n, d_ls)
        ctx.sync()
        assert d_u.numpy()[:len(u)].tobytes() == u
        assert d_lr.numpy().tobytes() == lr.tobytes()
        assert d_ls.numpy().tobytes() == ls.tobytes()
        assert rx_d.last_ms > 0 and tx_d.last_ms > 0
    for o in (rx_h, tx_h, rx_d, tx_d):
        o.close()


def test_iknp_matches_committed_golden(ctx, golden_dir):
    """the committed digests of tests/golden/stream_ot_golden.json (made with the oracle by make_golden_stream_ot.py)"""
    import importlib.util
    import json
    import os
    spec = importlib.util.spec_from_file_location("mk2", os.path.join(golden_dir, "make_golden_stream_ot.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = json.load(open(os.path.join(golden_dir, "stream_ot_golden.json")))
    for n in mk.IKNP_SIZES:
        base, delta, k0, b = mk.iknp_inputs(n)
        rcv, snd = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
        u, got = rcv.receive(b)
        sent = snd.send(u, n)
        assert mk.iknp_digest(u, got, sent) == gold["iknp"][str(n)]
        rcv.close(); snd.close()


def test_iknp_general_first_round(ctx, golden_dir, monkeypatch):
    """counters of 2^32 blocks and more (64 GiB of keystream per column) take the general first AES round, which no
    test can reach through the stream position: GC_IKNP_GENERIC=1 selects it for the same inputs"""
    monkeypatch.setenv("GC_IKNP_GENERIC", "1")
    test_iknp_matches_committed_golden(ctx, golden_dir)
    for n in (1, 513, 2049):
        test_iknp_matches_oracle_and_correlates(ctx, n)


def test_iknp_refuses_graph_capture(ctx):
    """the column streams advance with every call: inside gc_ctx_capture_* the device entry points return GC_E_ARG"""
    base, delta, k0 = base_setup("cap")
    rcv = engine.IKNPReceiver(ctx, base)
    d = ctx.zeros(8192 + 64 + 512 * 16)
    with pytest.raises(engine.EngineError) as e:
        ctx.capture(lambda: rcv.receive_dev(d, 512, d + 64, d + 64 + 8192))
    assert e.value.code == engine.GC_E_ARG
    u, got = rcv.receive(np.zeros(3, np.uint8))  # the pair is still usable and at position 0
    wu, wgot = oracle.IKNPReceiver(base).receive(np.zeros(3, np.uint8))
    assert bytes(u) == bytes(wu) and (got == wgot).all()
    rcv.close()


@pytest.mark.parametrize("n", [64, 1024, 1000, 70, 513, 4096 + 37])
def test_bitcot_device_resident_matches_host(ctx, n):
    """gc_iknp_*_bits_dev: choice words, u-matrix and result words stay in HBM; same bytes as the host calls (and so as
    the oracle's ReceiveBits / SendBits, incl. the whole-word-only choice fold), stream position persisting"""
    base, delta, k0 = base_setup("bitdev%d" % n)
    rx_h, tx_h = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    rx_d, tx_d = engine.IKNPReceiver(ctx, base), engine.IKNPSender(ctx, delta, k0)
    words = (n + 63) // 64
    for rep in range(2):
        choices = np.frombuffer(drbg("bcd%d/%d" % (n, rep), 8 * words), "<u8").copy()
        u, r = rx_h.receive_bits(choices, n)
        s = tx_h.send_bits(u, n)
        d_c = ctx.to_device(choices.view(np.uint8).copy())
        d_u = ctx.zeros(((n + 511) // 512) * 8192)
        d_r = ctx.zeros(words * 8)
        d_s = ctx.zeros(words * 8)
        rx_d.receive_bits_dev(d_c, n, d_u, d_r)
        tx_d.send_bits_dev(d_u, n, d_s)
        ctx.sync()
        assert d_u.numpy()[:len(u)].tobytes() == u
        assert (d_r.numpy().view("<u8") == r).all()
        assert (d_s.numpy().view("<u8") == s).all()
    for h in (rx_h, tx_h, rx_d, tx_d):
        h.close()


def test_cot_tuned_and_classic_kernels_agree(ctx, monkeypatch):
    """the dual-table persistent COT kernels (default) and the first, 4 KiB-table form (GC_COT_CLASSIC) are both the
    MITCCRH of ot/mitccrh.go: the reference's own vectors on each, ragged sizes"""
    import subprocess, sys, os
    code = ("import numpy as np, oracle; from mpc_amd import engine; from tests.test_gpu_ot import labels; from tests.util import drbg;"
            "ctx = engine.Context(0); seed = oracle.label_from_bytes(drbg('cls', 16)); x = labels('clsx', 3001);"
            "out = engine.mitccrh_hash(ctx, seed, 5, x, 1); m = oracle.MITCCRH(seed, 1);"
            "import hashlib; print(hashlib.sha256(out.tobytes()).hexdigest())")
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    a = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    b = subprocess.run([sys.executable, "-c", code], env=dict(env, GC_COT_CLASSIC="1"), capture_output=True, text=True, timeout=300)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert a.stdout.strip().splitlines()[-1] == b.stdout.strip().splitlines()[-1]


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, 4097])
def test_rot_pads_match_oracle_and_pair_up(ctx, n):
    """ROT (ot/rot.go:132-202): the sender's wires[j] become the two hashed pads, the receiver's result[j] the hashed pad of
    the label IKNP delivered; result[j] == wires[j].L{flag_j}; bytes == the oracle's restatement of both loops; the
    device-resident forms (labels straight from gc_iknp_*_dev) agree"""
    rng = np.random.default_rng(1000 + n)
    lab = lambda: (int(rng.integers(0, 1 << 63)), int(rng.integers(0, 1 << 63)))
    seed, delta = lab(), lab()
    data = np.zeros(n, LABEL)
    data["d0"] = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    data["d1"] = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    flags = rng.integers(0, 2, n).astype(bool)
    # what IKNP hands the receiver: the sender's label, ^ delta where the choice bit is set (iknp_test.go:98-113)
    recv = data.copy()
    recv["d0"][flags] ^= np.uint64(delta[0])
    recv["d1"][flags] ^= np.uint64(delta[1])
    wires = engine.rot_send(ctx, seed, delta, data)
    res = engine.rot_receive(ctx, seed, recv)
    assert wires.tobytes() == oracle.rot_send(seed, delta, data).tobytes()
    assert res.tobytes() == oracle.rot_receive(seed, recv).tobytes()
    want = np.where(flags, wires["l1"], wires["l0"]) if n else res
    assert (res == want).all(), "the receiver's pad is the sender's wires[j].L{flag}"
    if n:
        d_data, d_recv = ctx.to_device(data.view(np.uint8)), ctx.to_device(recv.view(np.uint8))
        d_w = ctx.zeros((n, 32))
        engine.rot_send_dev(ctx, seed, delta, d_data, n, d_w)
        engine.rot_receive_dev(ctx, seed, d_recv, n)
        assert d_w.numpy().tobytes() == wires.tobytes()
        assert d_recv.numpy().tobytes() == res.tobytes()


def test_rot_uses_the_reference_mitccrh_vectors(ctx):
    """the 8 vectors of ot/mitccrh_test.go:23-30 (seed 0, zero blocks, keys 0..7) through the ROT paths: a sender with zero
    IKNP labels and Delta = 0 gets them as both pads of wire j, a receiver with zero labels as result[j]"""
    z = np.zeros(8, LABEL)
    wires = engine.rot_send(ctx, (0, 0), (0, 0), z)
    res = engine.rot_receive(ctx, (0, 0), z)
    for i in range(8):
        assert oracle.label_to_bytes(wires[i]["l0"]).hex() == MITCCRH_BLOCKS[i]
        assert oracle.label_to_bytes(wires[i]["l1"]).hex() == MITCCRH_BLOCKS[i]
        assert oracle.label_to_bytes(res[i]).hex() == MITCCRH_BLOCKS[i]
