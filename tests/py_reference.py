"""A SECOND restatement of the reference's garbling path, independent of oracle/*.c: slow pure Python written from the
reference's Go text alone — ot/label.go:40-140 (Label, S, Mul2, Mul4, Xor, the BE(D0) || BE(D1) byte form),
circuit/garble.go:20-143 (idx, encrypt, decrypt, makeK, encryptHalf), :248-300 (Circuit.Garble: R, the input labels, the
slab in gate order), :311-482 (Gate.garbleInto), circuit/eval.go:17-115 (Circuit.Eval), circuit/stream_garble.go:41-192
(NewStreaming, Get / Set, Streaming.Garble) and :195-449 (garbleGate with its wire format), circuit/stream_evaluator.go:271-432
(the evaluator's loop over one OpCircuit block) — and AES itself from FIPS-197 (the reference calls Go's crypto/aes).

Why: the reference's own vectors pin AND / XOR and one OR / INV under a 32-byte key on one circuit (tests/go_transcript.py);
XNOR, 16- and 24-byte keys, OR / INV-dense circuits and the streaming wire format rest on ONE restatement of the Go text
(oracle/*.c) unless a second one, written separately, agrees with it (tests/test_py_reference.py).  Test
infrastructure only."""

XOR, XNOR, AND, OR, INV = 0, 1, 2, 3, 4
M64 = (1 << 64) - 1


# ---- AES (FIPS-197) ------------------------------------------------------------------------------------------------------
def _gmul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a = ((a << 1) ^ 0x11B) if a & 0x80 else a << 1
        b >>= 1
    return r & 0xFF


def _make_sbox():
    inv = [0] * 256
    for x in range(1, 256):  # multiplicative inverse: x^254
        y, e, p = 1, 254, x
        while e:
            if e & 1:
                y = _gmul(y, p)
            p = _gmul(p, p)
            e >>= 1
        inv[x] = y
    sbox = []
    for x in range(256):
        b = inv[x]
        s = b
        for k in range(1, 5):  # b ^ rotl(b,1) ^ rotl(b,2) ^ rotl(b,3) ^ rotl(b,4) ^ 0x63  (FIPS-197 5.1.1)
            s ^= ((b << k) | (b >> (8 - k))) & 0xFF
        sbox.append(s ^ 0x63)
    return sbox


SBOX = _make_sbox()


class AES:
    """aes.NewCipher(key).Encrypt: AES-128 / -192 / -256 of one 16-byte block"""

    def __init__(self, key):
        if len(key) not in (16, 24, 32):
            raise ValueError("crypto/aes: invalid key size %d" % len(key))
        nk = len(key) // 4
        self.nr = nk + 6
        w = [list(key[4 * i:4 * i + 4]) for i in range(nk)]
        rcon = 1
        for i in range(nk, 4 * (self.nr + 1)):  # FIPS-197 5.2
            t = list(w[i - 1])
            if i % nk == 0:
                t = [SBOX[t[1]] ^ rcon, SBOX[t[2]], SBOX[t[3]], SBOX[t[0]]]
                rcon = _gmul(rcon, 2)
            elif nk > 6 and i % nk == 4:
                t = [SBOX[b] for b in t]
            w.append([a ^ b for a, b in zip(w[i - nk], t)])
        self.rk = [sum((w[4 * r + c] for c in range(4)), []) for r in range(self.nr + 1)]

    def encrypt(self, block):
        s = [b ^ k for b, k in zip(block, self.rk[0])]  # state in column-major order, as the bytes come
        for r in range(1, self.nr + 1):
            s = [SBOX[b] for b in s]
            s = [s[(4 * c + 5 * i) % 16] for c in range(4) for i in range(4)]  # ShiftRows: row i of column c from column c + i
            if r != self.nr:
                m = []
                for c in range(4):
                    a = s[4 * c:4 * c + 4]
                    m += [_gmul(a[i], 2) ^ _gmul(a[(i + 1) % 4], 3) ^ a[(i + 2) % 4] ^ a[(i + 3) % 4] for i in range(4)]
                s = m
            s = [b ^ k for b, k in zip(s, self.rk[r])]
        return bytes(s)


# ---- ot.Label (label.go) ------------------------------------------------------------------------------------------------
def label_from_bytes(b):  # SetData: D0 = BE(b[0:8]), D1 = BE(b[8:16])
    return (int.from_bytes(b[0:8], "big"), int.from_bytes(b[8:16], "big"))


def label_bytes(l):  # GetData
    return l[0].to_bytes(8, "big") + l[1].to_bytes(8, "big")


def lxor(a, b):
    return (a[0] ^ b[0], a[1] ^ b[1])


def s_bit(l):
    return (l[0] >> 63) & 1


def set_s(l):
    return (l[0] | (1 << 63), l[1])


def mul2(l):
    return (((l[0] << 1) | (l[1] >> 63)) & M64, (l[1] << 1) & M64)


def mul4(l):
    return (((l[0] << 2) | (l[1] >> 62)) & M64, (l[1] << 2) & M64)


ZERO = (0, 0)


# ---- garble.go:20-143 --------------------------------------------------------------------------------------------------
def idx(l0, l1):
    return 2 * s_bit(l0) + s_bit(l1)


def make_k(a, b, t):
    return lxor(lxor(mul2(a), mul4(b)), (0, t))


def encrypt(alg, a, b, c, t):
    k = make_k(a, b, t)
    pi = label_from_bytes(alg.encrypt(label_bytes(k)))
    return lxor(lxor(pi, k), c)


def decrypt(alg, a, b, t, c):
    k = make_k(a, b, t)
    crypted = label_from_bytes(alg.encrypt(label_bytes(k)))
    return lxor(lxor(c, crypted), k)


def encrypt_half(alg, x, i):  # Hpi(x, i) = pi(K) ^ K, K = 2x ^ i
    k = lxor(mul2(x), (0, i))
    return lxor(label_from_bytes(alg.encrypt(label_bytes(k))), k)


def garble_gate(alg, r, op, a, b, idv):
    """garbleInto / garbleGate, the part both share: a, b wires (L0, L1); -> (c wire, table rows sent, new id)"""
    if op == XOR:
        l0 = lxor(a[0], b[0])
        return (l0, lxor(l0, r)), [], idv
    if op == XNOR:
        l0 = lxor(a[0], b[0])
        return (lxor(l0, r), l0), [], idv
    if op == AND:
        pa, pb = s_bit(a[0]), s_bit(b[0])
        j0, j1 = idv, idv + 1
        tg = lxor(encrypt_half(alg, a[0], j0), encrypt_half(alg, a[1], j0))
        if pb:
            tg = lxor(tg, r)
        wg0 = encrypt_half(alg, a[0], j0)
        if pa:
            wg0 = lxor(wg0, tg)
        te = lxor(lxor(encrypt_half(alg, b[0], j1), encrypt_half(alg, b[1], j1)), a[0])
        we0 = encrypt_half(alg, b[0], j1)
        if pb:
            we0 = lxor(lxor(we0, te), a[0])
        l0 = lxor(wg0, we0)
        return (l0, lxor(l0, r)), [tg, te], idv + 2
    if op == OR:
        table = [None] * 4
        c0 = c1 = ZERO  # (the output wire is still the zero wire when the rows are encrypted)
        table[idx(a[0], b[0])] = encrypt(alg, a[0], b[0], c0, idv)
        table[idx(a[0], b[1])] = encrypt(alg, a[0], b[1], c1, idv)
        table[idx(a[1], b[0])] = encrypt(alg, a[1], b[0], c1, idv)
        table[idx(a[1], b[1])] = encrypt(alg, a[1], b[1], c1, idv)
        l0i = idx(a[0], b[0])
        c0 = c1 = table[0]
        if l0i == 0:
            c1 = lxor(c1, r)
        else:
            c0 = lxor(c0, r)
        table = [lxor(t, c0 if i == l0i else c1) for i, t in enumerate(table)]
        return (c0, c1), table[1:4], idv + 1
    if op == INV:
        table = [None] * 2
        table[s_bit(a[0])] = encrypt(alg, a[0], ZERO, ZERO, idv)
        table[s_bit(a[1])] = encrypt(alg, a[1], ZERO, ZERO, idv)
        l0i = s_bit(a[0])
        c0 = c1 = table[0]
        if l0i == 0:
            c0 = lxor(c0, r)
        else:
            c1 = lxor(c1, r)
        table = [lxor(t, c1 if i == l0i else c0) for i, t in enumerate(table)]
        return (c0, c1), table[1:2], idv + 1
    raise ValueError("invalid gate type %d" % op)


def garble(gates, nwires, ninputs, key, rnd):
    """Circuit.Garble (garble.go:248-300): gates = [(in0, in1, out, op)]; rnd = the bytes the io.Reader delivers: R, then one L0
    per input wire.  -> (R, wires [(L0, L1)], rows per gate)"""
    r = set_s(label_from_bytes(rnd[0:16]))
    alg = AES(key)
    wires = [None] * nwires
    for i in range(ninputs):
        l0 = label_from_bytes(rnd[16 * (i + 1):16 * (i + 2)])
        wires[i] = (l0, lxor(l0, r))
    idv, tables = 0, []
    for in0, in1, out, op in gates:
        a = wires[in0]
        b = wires[in1] if op != INV else None
        wires[out], rows, idv = garble_gate(alg, r, op, a, b, idv)
        tables.append(rows)
    return r, wires, tables


def eval_gate(alg, op, a, b, rows, idv):
    """eval.go:28-112 / stream_evaluator.go:345-428: a, b labels -> (output label, new id)"""
    if op in (XOR, XNOR):
        return lxor(a, b), idv
    if op == AND:
        if len(rows) != 2:
            raise ValueError("corrupted ciruit: AND row length: %d" % len(rows))
        wg = encrypt_half(alg, a, idv)
        if s_bit(a):
            wg = lxor(wg, rows[0])
        we = encrypt_half(alg, b, idv + 1)
        if s_bit(b):
            we = lxor(lxor(we, rows[1]), a)
        return lxor(wg, we), idv + 2
    if op == OR:
        i = idx(a, b)
        c = rows[i - 1] if i > 0 else ZERO
        return decrypt(alg, a, b, idv, c), idv + 1
    if op == INV:
        i = s_bit(a)
        c = rows[i - 1] if i > 0 else ZERO
        return decrypt(alg, a, ZERO, idv, c), idv + 1
    raise ValueError("invalid operation %d" % op)


def evaluate(gates, key, wires, tables):
    """Circuit.Eval (eval.go:17-115): wires = labels, inputs filled in; in place"""
    alg = AES(key)
    idv = 0
    for (in0, in1, out, op), rows in zip(gates, tables):
        wires[out], idv = eval_gate(alg, op, wires[in0], wires[in1] if op != INV else ZERO, rows, idv)
    return wires


# ---- streaming (stream_garble.go, stream_evaluator.go) -------------------------------------------------------------------
class Stream:
    """NewStreaming + Streaming.Garble: the wire store, Get / Set through in[] / out[], the bytes of every gate"""

    def __init__(self, key, rnd, inputs):
        self.r = set_s(label_from_bytes(rnd[0:16]))
        self.alg = AES(key)
        self.wires = {}
        for i, w in enumerate(inputs):
            l0 = label_from_bytes(rnd[16 * (i + 1):16 * (i + 2)])
            self.wires[w] = (l0, lxor(l0, self.r))

    def wire(self, w):
        return self.wires.get(w, (ZERO, ZERO))  # (a wire nobody has set is the zero value of ot.Wire)

    def garble(self, gates, nwires, in_, out_):
        first_tmp, first_out = len(in_), nwires - len(out_)
        tmp = {}

        def get(w):  # Get (:131-141)
            if w < first_tmp:
                return self.wire(in_[w]), in_[w], False
            if w >= first_out:
                return self.wire(out_[w - first_out]), out_[w - first_out], False
            return tmp[w], w, True

        buf = bytearray()
        idv = 0
        for in0, in1, out, op in gates:
            b, bi, bt = get(in1) if op != INV else (None, 0, False)
            a, ai, at = get(in0)
            c, rows, idv = garble_gate(self.alg, self.r, op, a, b, idv)
            if out < first_tmp:  # (:385-394)
                ci, ct = in_[out], False
                self.wires[ci] = c
            elif out >= first_out:
                ci, ct = out_[out - first_out], False
                self.wires[ci] = c
            else:
                ci, ct = out, True
                tmp[out] = c
            opb = op | (0x80 if at else 0) | (0x40 if bt else 0) | (0x20 if ct else 0)
            ids = [ai, ci] if op == INV else [ai, bi, ci]
            if ai <= 0xFFFF and bi <= 0xFFFF and ci <= 0xFFFF:
                buf.append(opb | 0x10)
                for v in ids:
                    buf += v.to_bytes(2, "big")
            else:
                buf.append(opb)
                for v in ids:
                    buf += v.to_bytes(4, "big")
            for row in rows:
                buf += label_bytes(row)
        return bytes(buf)


class StreamEval:
    """the evaluator's loop over one OpCircuit block (stream_evaluator.go:271-432) and its two stores"""

    def __init__(self, key):
        self.alg = AES(key)
        self.wires = {}

    def set(self, w, label):
        self.wires[w] = label

    def get(self, w):
        return self.wires.get(w, ZERO)

    def circuit(self, ngates, data):
        tmp, pos, idv = {}, 0, 0
        for _ in range(ngates):
            gop = data[pos]
            pos += 1
            at, bt, ct, short = gop & 0x80, gop & 0x40, gop & 0x20, gop & 0x10
            op = gop & 0x0F
            if op > INV:
                raise ValueError("invalid operation %d" % op)
            sz = 2 if short else 4
            ids = []
            for _ in range(2 if op == INV else 3):
                ids.append(int.from_bytes(data[pos:pos + sz], "big"))
                pos += sz
            nrows = {XOR: 0, XNOR: 0, INV: 1, AND: 2, OR: 3}[op]
            rows = [label_from_bytes(data[pos + 16 * k:pos + 16 * k + 16]) for k in range(nrows)]
            pos += 16 * nrows
            a = tmp[ids[0]] if at else self.get(ids[0])
            b = ZERO if op == INV else (tmp[ids[1]] if bt else self.get(ids[1]))
            out, idv = eval_gate(self.alg, op, a, b, rows, idv)
            if ct:
                tmp[ids[-1]] = out
            else:
                self.wires[ids[-1]] = out
        return pos
