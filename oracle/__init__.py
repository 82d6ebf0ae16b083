"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the product package mpc_amd.  See
oracle/oracle.h for what is restated and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

LABEL = np.dtype([("d0", "<u8"), ("d1", "<u8")])
WIRE = np.dtype([("l0", LABEL), ("l1", LABEL)])
GATE = np.dtype(
    {
        "names": ["in0", "in1", "out", "op", "level"],
        "formats": ["<u4", "<u4", "<u4", "u1", "<u4"],
        "offsets": [0, 4, 8, 12, 16],
        "itemsize": 20,
    }
)
XOR, XNOR, AND, OR, INV = range(5)

E_KEYSIZE, E_RAND, E_GATE, E_ROWS, E_ARG = -1, -2, -3, -4, -5


class OracleError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s: oracle error %d" % (what, code))
        self.code = code


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    srcs = [os.path.join(HERE, f) for f in ("aes_oracle.c", "gc_oracle.c", "ot_oracle.c", "stream_oracle.c", "oracle.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return LIB_PATH


_lib = None


class _Aes(C.Structure):
    _fields_ = [("rk", C.c_uint8 * 240), ("rounds", C.c_int)]


class _Label(C.Structure):
    _fields_ = [("d0", C.c_uint64), ("d1", C.c_uint64)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_garble.restype = C.c_long
        L.orc_bench_garble_eval.restype = C.c_double
        L.orc_assign_levels.restype = C.c_uint32
        L.orc_iknp_receive.restype = C.c_size_t
        L.orc_iknp_send.restype = C.c_size_t
        L.orc_encrypt_half.restype = _Label
        L.orc_encrypt_half.argtypes = [C.c_void_p, _Label, C.c_uint32]
        L.orc_encrypt.restype = _Label
        L.orc_encrypt.argtypes = [C.c_void_p, _Label, _Label, _Label, C.c_uint32]
        L.orc_decrypt.restype = _Label
        L.orc_decrypt.argtypes = [C.c_void_p, _Label, _Label, C.c_uint32, _Label]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def _lab(x):
    """(d0, d1) tuple / numpy LABEL scalar -> _Label"""
    if isinstance(x, _Label):
        return x
    if isinstance(x, (tuple, list)):
        return _Label(int(x[0]), int(x[1]))
    return _Label(int(x["d0"]), int(x["d1"]))


# ---- AES ---------------------------------------------------------------------


def aes_encrypt(key, block, portable=False):
    L = lib()
    a = _Aes()
    k = _u8(key)
    rc = L.orc_aes_init(C.byref(a), _p(k), C.c_size_t(len(k)))
    if rc:
        raise OracleError(rc, "aes_init")
    i = _u8(block)
    o = np.zeros(16, np.uint8)
    L.orc_aes_force_portable(1 if portable else 0)
    L.orc_aes_encrypt(C.byref(a), _p(i), _p(o))
    L.orc_aes_force_portable(0)
    return o.tobytes()


def aes_round_keys(key):
    L = lib()
    a = _Aes()
    k = _u8(key)
    rc = L.orc_aes_init(C.byref(a), _p(k), C.c_size_t(len(k)))
    if rc:
        raise OracleError(rc, "aes_init")
    return bytes(a.rk)[: 16 * (a.rounds + 1)], a.rounds


def using_aesni():
    return bool(lib().orc_aes_using_aesni())


# ---- label helpers -------------------------------------------------------------


def label_from_bytes(b):
    """ot.Label.SetData: BE(D0) || BE(D1)"""
    return (int.from_bytes(b[0:8], "big"), int.from_bytes(b[8:16], "big"))


def label_to_bytes(l):
    l = _lab(l)
    return int(l.d0).to_bytes(8, "big") + int(l.d1).to_bytes(8, "big")


def label_mul2(l):
    x = _lab(l)
    lib().orc_label_mul2(C.byref(x))
    return (x.d0, x.d1)


def label_mul4(l):
    x = _lab(l)
    lib().orc_label_mul4(C.byref(x))
    return (x.d0, x.d1)


def label_set_s(l, on):
    x = _lab(l)
    lib().orc_label_set_s(C.byref(x), 1 if on else 0)
    return (x.d0, x.d1)


def label_bit(l, i):
    x = _lab(l)
    return int(lib().orc_label_bit(C.byref(x), int(i)))


def _aes_obj(key):
    a = _Aes()
    k = _u8(key)
    rc = lib().orc_aes_init(C.byref(a), _p(k), C.c_size_t(len(k)))
    if rc:
        raise OracleError(rc, "aes_init")
    return a


def encrypt_half(key, x, i):
    a = _aes_obj(key)
    r = lib().orc_encrypt_half(C.byref(a), _lab(x), C.c_uint32(i))
    return (r.d0, r.d1)


def encrypt(key, a_, b_, c_, t):
    a = _aes_obj(key)
    r = lib().orc_encrypt(C.byref(a), _lab(a_), _lab(b_), _lab(c_), C.c_uint32(t))
    return (r.d0, r.d1)


def decrypt(key, a_, b_, t, c_):
    a = _aes_obj(key)
    r = lib().orc_decrypt(C.byref(a), _lab(a_), _lab(b_), C.c_uint32(t), _lab(c_))
    return (r.d0, r.d1)


# ---- circuits ------------------------------------------------------------------


def slab_rows(gates):
    op = gates["op"]
    return int(2 * np.count_nonzero(op == AND) + 3 * np.count_nonzero(op == OR) + np.count_nonzero(op == INV))


def garble(gates, nwires, ninputs, key, rnd):
    """Circuit.Garble for one instance.  Returns dict(R, wires[WIRE], slab[LABEL], gate_off)."""
    L = lib()
    gates = np.ascontiguousarray(gates, dtype=GATE)
    k, r = _u8(key), _u8(rnd)
    cap = slab_rows(gates)
    wires = np.zeros(nwires, WIRE)
    slab = np.zeros(max(cap, 1), LABEL)
    goff = np.zeros(len(gates) + 1, np.uint32)
    R = np.zeros(1, LABEL)
    rows = L.orc_garble(_p(gates), C.c_uint32(len(gates)), C.c_uint32(nwires), C.c_uint32(ninputs), _p(k),
                        C.c_size_t(len(k)), _p(r), C.c_size_t(len(r)), _p(R), _p(wires), _p(slab),
                        C.c_size_t(cap), _p(goff))
    if rows < 0:
        raise OracleError(rows, "garble")
    return {"R": R[0], "wires": wires, "slab": slab[:rows], "gate_off": goff}


def eval_(gates, nwires, key, wires, slab):
    """Circuit.Eval: wires is a LABEL array of len nwires with inputs pre-filled (in place)."""
    L = lib()
    gates = np.ascontiguousarray(gates, dtype=GATE)
    k = _u8(key)
    slab = np.ascontiguousarray(slab, dtype=LABEL)
    assert wires.dtype == LABEL and len(wires) == nwires and wires.flags.c_contiguous
    rc = L.orc_eval(_p(gates), C.c_uint32(len(gates)), C.c_uint32(nwires), _p(k), C.c_size_t(len(k)), _p(wires),
                    _p(slab), C.c_size_t(len(slab)))
    if rc:
        raise OracleError(rc, "eval")
    return wires


def compute(gates, nwires, ninputs, in_bits):
    L = lib()
    gates = np.ascontiguousarray(gates, dtype=GATE)
    ib = np.ascontiguousarray(in_bits, dtype=np.uint8)
    assert len(ib) == ninputs
    wb = np.zeros(nwires, np.uint8)
    rc = L.orc_compute(_p(gates), C.c_uint32(len(gates)), C.c_uint32(nwires), C.c_uint32(ninputs), _p(ib), _p(wb))
    if rc:
        raise OracleError(rc, "compute")
    return wb


def assign_levels(gates, nwires):
    L = lib()
    g = np.ascontiguousarray(gates, dtype=GATE).copy()
    mw = C.c_uint32(0)
    nl = L.orc_assign_levels(_p(g), C.c_uint32(len(g)), C.c_uint32(nwires), C.byref(mw))
    return g, int(nl), int(mw.value)


def bench_garble_eval(gates, nwires, ninputs, noutputs, key, reps, threads):
    L = lib()
    gates = np.ascontiguousarray(gates, dtype=GATE)
    k = _u8(key)
    dt = L.orc_bench_garble_eval(_p(gates), C.c_uint32(len(gates)), C.c_uint32(nwires), C.c_uint32(ninputs),
                                 C.c_uint32(noutputs), _p(k), C.c_size_t(len(k)), C.c_uint32(reps), C.c_int(threads))
    if dt < 0:
        raise OracleError(-1, "bench_garble_eval")
    return float(dt)


# ---- OT --------------------------------------------------------------------------


class _Prg(C.Structure):
    _fields_ = [("aes", _Aes), ("ctr", C.c_uint8 * 16), ("ks", C.c_uint8 * 16), ("used", C.c_int)]


class Prg:
    """newPrg / prg / prgLabels (ot/iknp.go:622-645)"""

    def __init__(self, key):
        self.s = _Prg()
        lib().orc_prg_init(C.byref(self.s), _lab(key))

    def bytes(self, n):
        out = np.zeros(max(n, 1), np.uint8)
        lib().orc_prg_bytes(C.byref(self.s), _p(out), C.c_size_t(n))
        return out[:n].tobytes()

    def labels(self, n):
        out = np.zeros(max(n, 1), LABEL)
        lib().orc_prg_labels(C.byref(self.s), _p(out), C.c_size_t(n))
        return out[:n]


class _Receiver(C.Structure):
    _fields_ = [("g0", _Prg * 128), ("g1", _Prg * 128)]


class _Sender(C.Structure):
    _fields_ = [("g0", _Prg * 128), ("delta", _Label)]


def u_bytes(n):
    """bytes the receiver sends for n OTs (all chunks concatenated)"""
    total, ofs = 0, 0
    while ofs < n:
        rows = min(512, n - ofs)
        total += ((rows + 7) // 8) * 128
        ofs += rows
    return total


class IKNPReceiver:
    def __init__(self, base_wires):
        bw = np.ascontiguousarray(base_wires, dtype=WIRE)
        assert len(bw) == 128
        self.s = _Receiver()
        lib().orc_iknp_receiver_init(C.byref(self.s), _p(bw))

    def receive(self, b):
        """returns (u bytes, labels)"""
        bb = np.ascontiguousarray(b, dtype=np.uint8)
        n = len(bb)
        u = np.zeros(max(u_bytes(n), 1), np.uint8)
        res = np.zeros(max(n, 1), LABEL)
        w = lib().orc_iknp_receive(C.byref(self.s), _p(bb), C.c_size_t(n), _p(u), _p(res))
        return u[:w].tobytes(), res[:n]


class IKNPSender:
    def __init__(self, delta, k0):
        k = np.ascontiguousarray(k0, dtype=LABEL)
        assert len(k) == 128
        self.s = _Sender()
        lib().orc_iknp_sender_init(C.byref(self.s), _lab(delta), _p(k))

    def send(self, u, n):
        ub = _u8(u) if len(u) else np.zeros(1, np.uint8)
        res = np.zeros(max(n, 1), LABEL)
        lib().orc_iknp_send(C.byref(self.s), _p(ub), C.c_size_t(n), _p(res))
        return res[:n]


def create_labels(buf, w, nl):
    b = _u8(buf)
    out = np.zeros(max(nl, 1), LABEL)
    lib().orc_create_labels(_p(out), C.c_size_t(nl), _p(b), C.c_int(w))
    return out[:nl]


class _Mitccrh(C.Structure):
    _fields_ = [("batch_size", C.c_int), ("start", _Label), ("gid", C.c_uint64), ("ciphers", _Aes * 8),
                ("key_used", C.c_int)]


class MITCCRH:
    def __init__(self, seed, batch_size=8):
        self.s = _Mitccrh()
        lib().orc_mitccrh_init(C.byref(self.s), _lab(seed), C.c_int(batch_size))

    def hash(self, blks, k, h):
        assert blks.dtype == LABEL and len(blks) == k * h
        lib().orc_mitccrh_hash(C.byref(self.s), _p(blks), C.c_int(k), C.c_int(h))
        return blks


def cot_send_pads(seed, delta, data, wires):
    d = np.ascontiguousarray(data, dtype=LABEL)
    w = np.ascontiguousarray(wires, dtype=WIRE)
    n = len(d)
    out = np.zeros(max(2 * n, 1), LABEL)
    lib().orc_cot_send_pads(_lab(seed), _lab(delta), _p(d), _p(w), C.c_size_t(n), _p(out))
    return out[: 2 * n]


def rot_send(seed, delta, data):
    """ROT.Send pad loop (ot/rot.go:156-172): WIRE[n] = {H_j(data_j), H_j(data_j ^ delta)}"""
    d = np.ascontiguousarray(data, dtype=LABEL)
    n = len(d)
    out = np.zeros(max(n, 1), WIRE)
    lib().orc_rot_send(_lab(seed), _lab(delta), _p(d), C.c_size_t(n), _p(out))
    return out[:n]


def rot_receive(seed, result):
    """ROT.Receive pad loop (ot/rot.go:194-199): the hashed pads of the labels IKNP delivered"""
    r = np.ascontiguousarray(result, dtype=LABEL).copy()
    lib().orc_rot_receive(_lab(seed), _p(r), C.c_size_t(len(r)))
    return r


def cot_receive_unpad(seed, flags, sent, result):
    f = np.ascontiguousarray(flags, dtype=np.uint8)
    s = np.ascontiguousarray(sent, dtype=LABEL)
    r = np.ascontiguousarray(result, dtype=LABEL).copy()
    lib().orc_cot_receive_unpad(_lab(seed), _p(f), _p(s), _p(r), C.c_size_t(len(f)))
    return r


def mul128(a, b, ref=False):
    lo, hi = _Label(), _Label()
    fn = lib().orc_mul128_ref if ref else lib().orc_mul128
    fn.argtypes = [_Label, _Label, C.c_void_p, C.c_void_p]
    fn(_lab(a), _lab(b), C.byref(lo), C.byref(hi))
    return (lo.d0, lo.d1), (hi.d0, hi.d1)


def inner_product(a, b):
    a = np.ascontiguousarray(a, dtype=LABEL)
    b = np.ascontiguousarray(b, dtype=LABEL)
    r1, r2 = _Label(), _Label()
    lib().orc_inner_product(_p(a), _p(b), C.c_size_t(min(len(a), len(b))), C.byref(r1), C.byref(r2))
    return (r1.d0, r1.d1), (r2.d0, r2.d1)


# ---- streaming garbler / evaluator ----------------------------------------------------------


class Stream:
    """circuit.Streaming (stream_garble.go): NewStreaming + Garble + GetInput"""

    def __init__(self, key, rnd, inputs):
        L = lib()
        L.orc_stream_new.restype = C.c_void_p
        L.orc_stream_garble.restype = C.c_long
        k, r = _u8(key), _u8(rnd)
        inp = np.ascontiguousarray(inputs, dtype=np.uint32)
        st = C.c_int(0)
        self.h = L.orc_stream_new(_p(k), C.c_size_t(len(k)), _p(r), C.c_size_t(len(r)), _p(inp), C.c_uint32(len(inp)),
                                  C.byref(st))
        if not self.h:
            raise OracleError(st.value, "stream_new")

    def get(self, w):
        out = np.zeros(1, WIRE)
        rc = lib().orc_stream_get(C.c_void_p(self.h), C.c_uint32(w), _p(out))
        if rc:
            raise OracleError(rc, "stream_get")
        return out[0]

    def garble(self, gates, nwires, in_, out_):
        g = np.ascontiguousarray(gates, dtype=GATE)
        i = np.ascontiguousarray(in_, dtype=np.uint32)
        o = np.ascontiguousarray(out_, dtype=np.uint32)
        buf = np.zeros(len(g) * 61 + 16, np.uint8)
        n = lib().orc_stream_garble(C.c_void_p(self.h), _p(g), C.c_uint32(len(g)), C.c_uint32(nwires), _p(i),
                                    C.c_uint32(len(i)), _p(o), C.c_uint32(len(o)), _p(buf), C.c_size_t(len(buf)))
        if n < 0:
            raise OracleError(n, "stream_garble")
        return buf[:n].tobytes()

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_stream_free(C.c_void_p(self.h))
            self.h = None


class StreamEval:
    """StreamEval store + the per-gate loop of stream_evaluator.go:271-432"""

    def __init__(self, key):
        L = lib()
        L.orc_stream_eval_new.restype = C.c_void_p
        L.orc_stream_eval_circuit.restype = C.c_long
        k = _u8(key)
        st = C.c_int(0)
        self.h = L.orc_stream_eval_new(_p(k), C.c_size_t(len(k)), C.byref(st))
        if not self.h:
            raise OracleError(st.value, "stream_eval_new")

    def set(self, w, label):
        lib().orc_stream_eval_set.argtypes = [C.c_void_p, C.c_uint32, _Label]
        rc = lib().orc_stream_eval_set(C.c_void_p(self.h), C.c_uint32(w), _lab(label))
        if rc:
            raise OracleError(rc, "stream_eval_set")

    def get(self, w):
        out = _Label()
        rc = lib().orc_stream_eval_get(C.c_void_p(self.h), C.c_uint32(w), C.byref(out))
        if rc:
            raise OracleError(rc, "stream_eval_get")
        return (out.d0, out.d1)

    def circuit(self, ngates, ntmp, nwires, data):
        b = _u8(data) if len(data) else np.zeros(1, np.uint8)
        n = lib().orc_stream_eval_circuit(C.c_void_p(self.h), C.c_uint32(ngates), C.c_uint32(ntmp), C.c_uint32(nwires),
                                          _p(b), C.c_size_t(len(data)))
        if n < 0:
            raise OracleError(n, "stream_eval_circuit")
        return n

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_stream_eval_free(C.c_void_p(self.h))
            self.h = None


# ---- table wire format ---------------------------------------------------------------------------


def tables_serialize(gates, slab):
    g = np.ascontiguousarray(gates, dtype=GATE)
    sl = np.ascontiguousarray(slab, dtype=LABEL)
    out = np.zeros(4 + 4 * len(g) + 16 * len(sl), np.uint8)
    lib().orc_tables_serialize.restype = C.c_size_t
    n = lib().orc_tables_serialize(_p(g), C.c_uint32(len(g)), _p(sl) if len(sl) else None, _p(out))
    return out[:n].tobytes()


def tables_parse(gates, data):
    g = np.ascontiguousarray(gates, dtype=GATE)
    b = _u8(data)
    slab = np.zeros(max(slab_rows(g), 1), LABEL)
    lib().orc_tables_parse.restype = C.c_long
    n = lib().orc_tables_parse(_p(g), C.c_uint32(len(g)), _p(b), C.c_size_t(len(b)), _p(slab))
    if n < 0:
        raise OracleError(n, "tables_parse")
    return slab[:n]


# ---- KOS consistency check ---------------------------------------------------------------------------


def kos_receiver_tags(seed2, result, b, choice_vec, bcv):
    r = np.ascontiguousarray(result, dtype=LABEL)
    cv = np.ascontiguousarray(choice_vec, dtype=LABEL)
    bb = np.ascontiguousarray(b, dtype=np.uint8)
    bc = np.ascontiguousarray(bcv, dtype=np.uint8)
    assert len(cv) == 256 and len(bc) == 256
    x, t0, t1 = _Label(), _Label(), _Label()
    if len(r) == 0:
        r = np.zeros(1, LABEL)
        bb = np.zeros(1, np.uint8)
        n = 0
    else:
        n = len(result)
    lib().orc_kos_receiver_tags(_lab(seed2), _p(r), _p(bb), C.c_size_t(n), _p(cv), _p(bc), C.byref(x), C.byref(t0),
                                C.byref(t1))
    return (x.d0, x.d1), (t0.d0, t0.d1), (t1.d0, t1.d1)


def kos_sender_check(seed2, result, choice_vec, delta, x, t0, t1):
    r = np.ascontiguousarray(result, dtype=LABEL)
    cv = np.ascontiguousarray(choice_vec, dtype=LABEL)
    n = len(r)
    if n == 0:
        r = np.zeros(1, LABEL)
    lib().orc_kos_sender_check.argtypes = [_Label, C.c_void_p, C.c_size_t, C.c_void_p, _Label, _Label, _Label, _Label]
    return bool(lib().orc_kos_sender_check(_lab(seed2), _p(r), C.c_size_t(n), _p(cv), _lab(delta), _lab(x), _lab(t0),
                                           _lab(t1)))


def iknp_receive_bits(rcv, choices, n):
    """ReceiveBits (iknp.go:554-620); rcv is an oracle.IKNPReceiver"""
    ch = np.ascontiguousarray(choices, dtype=np.uint64)
    u = np.zeros(max(u_bytes(n), 1), np.uint8)
    res = np.zeros(max((n + 63) // 64, 1), np.uint64)
    lib().orc_iknp_receive_bits.restype = C.c_size_t
    w = lib().orc_iknp_receive_bits(C.byref(rcv.s), _p(ch), C.c_size_t(n), _p(u), _p(res))
    return u[:w].tobytes(), res[: (n + 63) // 64]


def iknp_send_bits(snd, u, n):
    """SendBits (iknp.go:259-310); snd is an oracle.IKNPSender"""
    ub = _u8(u) if len(u) else np.zeros(1, np.uint8)
    res = np.zeros(max((n + 63) // 64, 1), np.uint64)
    lib().orc_iknp_send_bits.restype = C.c_size_t
    lib().orc_iknp_send_bits(C.byref(snd.s), _p(ub), C.c_size_t(n), _p(res))
    return res[: (n + 63) // 64]
