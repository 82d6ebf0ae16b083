"""Oracle IKNP / COT: the properties the reference's own tests assert
(ot/iknp_test.go:98-113 correlation, ot/ot_test.go:83-97 delivery), incl. its chunk-size cases."""
import numpy as np
import pytest

import oracle
from tests.util import drbg


def labels(seed, n):
    raw = drbg(seed, 16 * n)
    out = np.zeros(n, oracle.LABEL)
    for i in range(n):
        out[i] = oracle.label_from_bytes(raw[16 * i : 16 * i + 16])
    return out


def setup(seed):
    """base-OT outcome: receiver holds (L0,L1) x128, sender holds L_{delta bit i}"""
    base = np.zeros(128, oracle.WIRE)
    base["l0"] = labels(seed + "l0", 128)
    base["l1"] = labels(seed + "l1", 128)
    delta = oracle.label_from_bytes(drbg(seed + "delta", 16))
    k0 = np.zeros(128, oracle.LABEL)
    for i in range(128):
        k0[i] = base[i]["l1"] if oracle.label_bit(delta, i) else base[i]["l0"]
    return oracle.IKNPReceiver(base), oracle.IKNPSender(delta, k0), delta


# iknp_test.go:32-37 chunk-size values (the Go loop iterates indices 0..9; both sets are covered)
SIZES = [0, 1, 2, 3, 7, 8, 9, 129, 512, 513, 1024, 1025, 1536, 1537, 2048, 2049, 2560, 700]


@pytest.mark.parametrize("n", SIZES)
def test_iknp_correlation(n):
    rcv, snd, delta = setup("iknp%d" % n)
    b = np.frombuffer(drbg("b%d" % n, max(n, 1)), np.uint8)[:n] & 1
    u, got = rcv.receive(b)
    assert len(u) == oracle.u_bytes(n)
    sent = snd.send(u, n)
    for i in range(n):  # rcvd[i] == sent[i] ^ b[i]*delta
        want = (int(sent[i]["d0"]) ^ (delta[0] if b[i] else 0), int(sent[i]["d1"]) ^ (delta[1] if b[i] else 0))
        assert (int(got[i]["d0"]), int(got[i]["d1"])) == want


def test_iknp_streams_persist_across_calls():
    # the per-column CTR streams continue across Send/Receive calls, also from a mid-block offset
    rcv, snd, delta = setup("persist")
    for n in (24 * 8, 5, 1000):  # byteRows 24 -> the next call starts mid AES block
        b = np.frombuffer(drbg("pb%d" % n, n), np.uint8) & 1
        u, got = rcv.receive(b)
        sent = snd.send(u, n)
        x0 = sent["d0"] ^ np.where(b == 1, np.uint64(delta[0]), np.uint64(0))
        x1 = sent["d1"] ^ np.where(b == 1, np.uint64(delta[1]), np.uint64(0))
        assert (got["d0"] == x0).all() and (got["d1"] == x1).all()


def test_create_labels_bit_order():
    # label r = 8*row + bit has bit j = bit `bit` of buf[j*w + row]; j<64 -> D0 (iknp.go:647-683)
    w = 3
    buf = bytearray(128 * w)
    buf[5 * w + 1] = 0x04  # column 5, row-byte 1, bit 2 -> label 10, bit 5 (D0)
    buf[100 * w + 2] = 0x80  # column 100, row-byte 2, bit 7 -> label 23, bit 36 of D1
    out = oracle.create_labels(bytes(buf), w, 24)
    assert int(out[10]["d0"]) == 1 << 5 and int(out[23]["d1"]) == 1 << 36
    assert sum(int(x["d0"] != 0) + int(x["d1"] != 0) for x in out) == 2
    # truncation when the label slice is shorter than 8*w
    assert len(oracle.create_labels(bytes(buf), w, 11)) == 11


@pytest.mark.parametrize("n", [1, 8, 13, 64])
def test_cot_delivers_chosen_label(n):
    # ot_test.go:83-97: alternating flags, receiver obtains wires[i].L{flag}
    rcv, snd, delta = setup("cot%d" % n)
    flags = np.array([i % 2 for i in range(n)], np.uint8)
    wires = np.zeros(n, oracle.WIRE)
    wires["l0"] = labels("w0", n)
    wires["l1"] = labels("w1", n)
    u, got = rcv.receive(flags)
    data = snd.send(u, n)
    seed = oracle.label_from_bytes(drbg("cotseed", 16))
    sent = oracle.cot_send_pads(seed, delta, data, wires)
    res = oracle.cot_receive_unpad(seed, flags, sent, got)
    for i in range(n):
        assert res[i] == (wires[i]["l1"] if flags[i] else wires[i]["l0"])


def test_kos_check_accepts_honest_rejects_tampered():
    # malicious variant (iknp.go:138-194, 373-465): the sender's check passes on an honest run and fails when
    # one receiver label is flipped ("OT extension check failed")
    rcv, snd, delta = setup("kos")
    n = 700
    b = np.frombuffer(drbg("kb", n), np.uint8) & 1
    u, got = rcv.receive(b)
    sent = snd.send(u, n)
    bcv = np.frombuffer(drbg("kbcv", 256), np.uint8) & 1
    u2, cvr = rcv.receive(bcv)
    cvs = snd.send(u2, 256)
    seed2 = oracle.label_from_bytes(drbg("seed2", 16))
    x, t0, t1 = oracle.kos_receiver_tags(seed2, got, b, cvr, bcv)
    assert oracle.kos_sender_check(seed2, sent, cvs, delta, x, t0, t1)
    bad = got.copy()
    bad[5]["d0"] ^= 1
    x2, t0b, t1b = oracle.kos_receiver_tags(seed2, bad, b, cvr, bcv)
    assert not oracle.kos_sender_check(seed2, sent, cvs, delta, x2, t0b, t1b)
    # x is the XOR of the chi labels selected by the choice bits
    chi = oracle.Prg(seed2).labels(n + 256)
    sel = np.concatenate([b, bcv]).astype(bool)
    assert x == (int(np.bitwise_xor.reduce(chi["d0"][sel])), int(np.bitwise_xor.reduce(chi["d1"][sel])))


def test_bitcot_correlation():
    # ot/bitcot_test.go:14-87: s ^ r == Delta.Bit(0) & c  for 1024 bits (multiples of 64 — the reference folds
    # whole choice words only)
    rcv, snd, delta = setup("bitcot")
    n = 1024
    choices = np.frombuffer(drbg("bc", n // 8), "<u8").copy()
    u, r = oracle.iknp_receive_bits(rcv, choices, n)
    s = oracle.iknp_send_bits(snd, u, n)
    d0 = np.uint64(0xFFFFFFFFFFFFFFFF) if oracle.label_bit(delta, 0) else np.uint64(0)
    assert ((s ^ r) == (choices & d0)).all()


def test_rot_pads_pair_up_and_hit_the_reference_vectors():
    """oracle restatement of ot/rot.go:156-172,194-199: result[j] == wires[j].L{flag_j}; with zero labels, zero Delta and
    seed 0 both loops reproduce ot/mitccrh_test.go:23-30"""
    import numpy as np
    from tests.test_oracle_kat import MITCCRH_BLOCKS
    rng = np.random.default_rng(5)
    for n in (1, 8, 9, 100):
        seed = (int(rng.integers(0, 1 << 63)), int(rng.integers(0, 1 << 63)))
        delta = (int(rng.integers(0, 1 << 63)), int(rng.integers(0, 1 << 63)))
        data = np.zeros(n, oracle.LABEL)
        data["d0"] = rng.integers(0, 1 << 63, n, dtype=np.uint64)
        data["d1"] = rng.integers(0, 1 << 63, n, dtype=np.uint64)
        flags = rng.integers(0, 2, n).astype(bool)
        recv = data.copy()
        recv["d0"][flags] ^= np.uint64(delta[0])
        recv["d1"][flags] ^= np.uint64(delta[1])
        wires = oracle.rot_send(seed, delta, data)
        res = oracle.rot_receive(seed, recv)
        assert (res == np.where(flags, wires["l1"], wires["l0"])).all()
        # the same through the MITCCRH object directly (mitccrh.go:93-128): OT j under key index j
        m = oracle.MITCCRH(seed, 8)
        for i in range(0, n, 8):
            k = min(8, n - i)
            pad = np.zeros(8, oracle.LABEL)
            pad[:k] = recv[i:i + k]
            m.hash(pad, 8, 1)
            assert (pad[:k] == res[i:i + k]).all()
    z = np.zeros(8, oracle.LABEL)
    w = oracle.rot_send((0, 0), (0, 0), z)
    r = oracle.rot_receive((0, 0), z)
    for i in range(8):
        assert oracle.label_to_bytes(w[i]["l0"]).hex() == MITCCRH_BLOCKS[i] == oracle.label_to_bytes(r[i]).hex()
