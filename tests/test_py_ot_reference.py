"""Two restatements of the reference's OT-extension path that were written apart — oracle/ot_oracle.c (C) and
tests/py_ot_reference.py (Python, from the Go text alone) — agree on every byte: the u-matrix the receiver sends, both sides'
labels, the state of the column PRGs across calls, the COT / ROT pads, the KOS tags and the packed bit-COT words.  No
reference-held vector reaches these (the reference tests them through their properties, ot/iknp_test.go:98-113); the MITCCRH
keys, which every pad goes through, are also pinned by the reference's own vectors (ot/mitccrh_test.go:23-30), and the Python
side reproduces those here too."""
import numpy as np
import pytest

import oracle
from tests import py_ot_reference as po
from tests.util import drbg


def lab(seed):
    return oracle.label_from_bytes(drbg(seed, 16))


def labs(seed, n):
    raw = drbg(seed, 16 * max(n, 1))
    return [oracle.label_from_bytes(raw[16 * i:16 * i + 16]) for i in range(n)]


def np_labels(ls):
    out = np.zeros(len(ls), oracle.LABEL)
    for i, l in enumerate(ls):
        out[i] = l
    return out


def tup(arr):
    return [(int(x["d0"]), int(x["d1"])) for x in arr]


def both(seed):
    l0, l1 = labs(seed + "l0", 128), labs(seed + "l1", 128)
    delta = lab(seed + "delta")
    base = np.zeros(128, oracle.WIRE)
    base["l0"], base["l1"] = np_labels(l0), np_labels(l1)
    k0 = [l1[i] if po.bit(delta, i) else l0[i] for i in range(128)]
    return (oracle.IKNPReceiver(base), oracle.IKNPSender(delta, np_labels(k0)),
            po.Receiver(list(zip(l0, l1))), po.Sender(delta, k0), delta)


def test_prg_is_ctr_mode_and_persists():
    key = lab("prgkey")
    a, b = oracle.Prg(key), po.Prg(key)
    for n in (0, 1, 15, 16, 17, 64, 5, 100):  # (calls that end and start in the middle of a key-stream block)
        assert a.bytes(n) == b.bytes(n)
    assert tup(a.labels(5)) == b.labels(5)


# ot/iknp_test.go:32-37's sizes (chunk boundaries at 512 rows) and ragged ones
@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 129, 512, 513, 700, 1025])
def test_iknp_expansion_bytes_agree(n):
    orc_r, orc_s, py_r, py_s, delta = both("x%d" % n)
    b = [int(v) & 1 for v in drbg("b%d" % n, max(n, 1))[:n]]
    u_o, got_o = orc_r.receive(np.array(b, np.uint8))
    u_p, got_p = py_r.receive(b)
    assert u_o == u_p
    assert tup(got_o) == got_p
    sent_o = orc_s.send(u_o, n)
    sent_p = py_s.send(u_p, n)
    assert tup(sent_o) == sent_p
    for i in range(n):  # ... and the property the reference tests (iknp_test.go:98-113)
        assert got_p[i] == (po.lxor(sent_p[i], delta) if b[i] else sent_p[i])


def test_iknp_streams_continue_across_calls():
    orc_r, orc_s, py_r, py_s, delta = both("persist")
    for n in (24 * 8, 5, 600):  # 24 byte rows: the next call starts in the middle of an AES block
        b = [int(v) & 1 for v in drbg("pb%d" % n, n)]
        u_o, got_o = orc_r.receive(np.array(b, np.uint8))
        u_p, got_p = py_r.receive(b)
        assert u_o == u_p and tup(got_o) == got_p
        assert tup(orc_s.send(u_o, n)) == py_s.send(u_p, n)


def test_create_labels_agree():
    for w, nl in ((3, 24), (3, 11), (1, 8), (64, 512), (5, 33)):
        buf = drbg("cl%d" % w, 128 * w)
        assert tup(oracle.create_labels(buf, w, nl)) == po.create_labels(nl, buf, w)


def test_mitccrh_reference_vectors_and_agreement():
    from tests.test_oracle_kat import MITCCRH_BLOCKS  # ot/mitccrh_test.go:23-30
    m = po.Mitccrh((0, 0), 8)
    got = m.hash([(0, 0)] * 8, 8, 1)
    assert [po.label_bytes(x).hex() for x in got] == MITCCRH_BLOCKS
    seed = lab("mseed")
    mo, mp = oracle.MITCCRH(seed, 8), po.Mitccrh(seed, 8)
    for k, h in ((8, 2), (8, 1), (4, 3), (2, 1), (2, 1), (8, 2)):  # (key batches renewed in the middle of the sequence)
        blks = labs("blk%d%d" % (k, h), k * h)
        assert tup(mo.hash(np_labels(blks), k, h)) == mp.hash(blks, k, h)


@pytest.mark.parametrize("n", [1, 8, 13, 64, 100])
def test_cot_and_rot_pads_agree(n):
    orc_r, orc_s, py_r, py_s, delta = both("cot%d" % n)
    flags = [int(v) & 1 for v in drbg("cf%d" % n, n)]
    w0, w1 = labs("w0%d" % n, n), labs("w1%d" % n, n)
    u, got = py_r.receive(flags)
    data = py_s.send(u, n)
    seed = lab("cotseed%d" % n)
    wires = np.zeros(n, oracle.WIRE)
    wires["l0"], wires["l1"] = np_labels(w0), np_labels(w1)
    sent_o = oracle.cot_send_pads(seed, delta, np_labels(data), wires)
    sent_p = po.cot_send_pads(seed, delta, data, list(zip(w0, w1)))
    assert tup(sent_o) == sent_p
    res_o = oracle.cot_receive_unpad(seed, np.array(flags, np.uint8), sent_o, np_labels(got))
    res_p = po.cot_receive_unpad(seed, flags, sent_p, got)
    assert tup(res_o) == res_p
    assert res_p == [w1[i] if flags[i] else w0[i] for i in range(n)]  # ot_test.go:83-97
    rw_o = oracle.rot_send(seed, delta, np_labels(data))
    rw_p = po.rot_send(seed, delta, data)
    assert [(tuple(int(v) for v in x["l0"]), tuple(int(v) for v in x["l1"])) for x in rw_o] == rw_p
    rr_o = oracle.rot_receive(seed, np_labels(got))
    rr_p = po.rot_receive(seed, got)
    assert tup(rr_o) == rr_p
    assert rr_p == [rw_p[i][flags[i]] for i in range(n)]


def test_mul128_agrees():
    for i in range(20):
        a, b = lab("ma%d" % i), lab("mb%d" % i)
        assert oracle.mul128(a, b) == po.mul128(a, b)
    assert po.mul128((1, 0), (5, 7)) == ((5, 7), (0, 0))


@pytest.mark.parametrize("n", [0, 1, 700, 1024, 1500])
def test_kos_tags_agree(n):
    orc_r, orc_s, py_r, py_s, delta = both("kos%d" % n)
    b = [int(v) & 1 for v in drbg("kb%d" % n, max(n, 1))[:n]]
    u, got = py_r.receive(b)
    sent = py_s.send(u, n)
    bcv = [int(v) & 1 for v in drbg("kbcv%d" % n, 256)]
    u2, cvr = py_r.receive(bcv)
    cvs = py_s.send(u2, 256)
    seed2 = lab("seed2%d" % n)
    tags_p = po.kos_receiver_tags(seed2, got, b, cvr, bcv)
    tags_o = oracle.kos_receiver_tags(seed2, np_labels(got), np.array(b, np.uint8), np_labels(cvr), np.array(bcv, np.uint8))
    assert tags_o == tags_p
    assert po.kos_sender_check(seed2, sent, cvs, delta, *tags_p)
    assert oracle.kos_sender_check(seed2, np_labels(sent), np_labels(cvs), delta, *tags_p)
    if n:
        bad = list(got)
        bad[n // 2] = (bad[n // 2][0] ^ 1, bad[n // 2][1])
        tags_bad = po.kos_receiver_tags(seed2, bad, b, cvr, bcv)
        assert not po.kos_sender_check(seed2, sent, cvs, delta, *tags_bad)  # "OT extension check failed" (iknp.go:190)


@pytest.mark.parametrize("n", [64, 512, 1024, 576, 100])
def test_bitcot_words_agree(n):
    """incl. sizes that are no multiple of 64: the reference folds the choice bits in by whole words of a chunk's byte rows
    (`words := byteRows / 8`, iknp.go:576) — both restatements keep that"""
    orc_r, orc_s, py_r, py_s, delta = both("bit%d" % n)
    nw = (n + 63) // 64
    choices = [int.from_bytes(drbg("bc%d" % n, 8 * nw)[8 * i:8 * i + 8], "little") for i in range(nw)]
    u_o, r_o = oracle.iknp_receive_bits(orc_r, np.array(choices, np.uint64), n)
    u_p, r_p = py_r.receive_bits(choices, n)
    assert u_o == u_p
    assert [int(v) for v in r_o] == r_p
    s_o = oracle.iknp_send_bits(orc_s, u_o, n)
    s_p = py_s.send_bits(u_p, n)
    assert [int(v) for v in s_o] == s_p
    if n % 64 == 0:  # ot/bitcot_test.go:14-87: s ^ r == Delta.Bit(0) & c
        d0 = (1 << 64) - 1 if po.bit(delta, 0) else 0
        assert [a ^ b for a, b in zip(s_p, r_p)] == [c & d0 for c in choices]
