"""A SECOND restatement of the reference's OT-extension path, in plain Python, written from the Go text alone — ot/iknp.go,
ot/mitccrh.go, ot/cot.go, ot/rot.go, ot/gf128.go, ot/mul128_generic.go, ot/label.go — and NOT from oracle/ot_oracle.c.  Test
infrastructure (tests/test_py_ot_reference.py): where no reference-held vector reaches (the u-matrix and label bytes of the IKNP
expansion, the COT / ROT pads, the KOS tags, the packed bit-COT words) two restatements written apart have to agree byte for
byte; the MITCCRH keys are also pinned by the reference's own vectors (ot/mitccrh_test.go:23-30).  Slow on purpose: integers
and byte strings, its own AES (FIPS-197, tests/py_reference.py) and CTR mode (SP 800-38A)."""
from tests.py_reference import AES, label_bytes, label_from_bytes, lxor

K = 128                      # iknp.go:49
CHUNK_SIZE = 8 * 1024        # iknp.go:52
CHUNK_BYTE_ROWS = CHUNK_SIZE // K   # iknp.go:55
CHUNK_ROWS = CHUNK_BYTE_ROWS * 8    # iknp.go:58
OT_BATCH = 8                 # cot.go:47
M64 = (1 << 64) - 1


def bit(l, i):               # Label.Bit (label.go:129-141)
    return ((l[1] >> (i - 64)) if i > 63 else (l[0] >> i)) & 1


class Prg:
    """newPrg / prg (iknp.go:622-637): AES-CTR under the label's bytes, zero IV; the stream persists across calls
    (crypto/cipher.NewCTR: the 16-byte counter is one big-endian integer, a key-stream block is used up byte by byte)"""

    def __init__(self, key_label):
        self.aes = AES(label_bytes(key_label))
        self.ctr = 0
        self.left = b""

    def bytes(self, n):
        out = bytearray()
        while len(out) < n:
            if not self.left:
                self.left = self.aes.encrypt(self.ctr.to_bytes(16, "big"))
                self.ctr = (self.ctr + 1) & ((1 << 128) - 1)
            take = min(n - len(out), len(self.left))
            out += self.left[:take]
            self.left = self.left[take:]
        return bytes(out)

    def labels(self, n):     # prgLabels (iknp.go:639-645): SetBytes of 16 stream bytes each
        return [label_from_bytes(self.bytes(16)) for _ in range(n)]


def create_labels(nl, buf, w):
    """createLabels (iknp.go:647-683): label 8*row + b has bit j = bit b of buf[j*w + row]; at most nl labels"""
    out = []
    for row in range(w):
        eight = [[0, 0] for _ in range(8)]
        for j in range(128):
            byte = buf[j * w + row]
            for b in range(8):
                if (byte >> b) & 1:
                    eight[b][0 if j < 64 else 1] |= 1 << (j & 63)
        for b in range(8):
            if len(out) >= min(w * 8, nl):
                return out
            out.append((eight[b][0], eight[b][1]))
    return out


def _xor_prefix(dst, src):   # xor (co_helpers.go:238-249): over the shorter of the two
    n = min(len(dst), len(src))
    return bytes(a ^ b for a, b in zip(dst[:n], src[:n])) + dst[n:]


class Receiver:
    """IKNPReceiver (iknp.go:331-356): two PRGs per base-OT wire"""

    def __init__(self, base_wires):
        assert len(base_wires) == K
        self.g0 = [Prg(w[0]) for w in base_wires]
        self.g1 = [Prg(w[1]) for w in base_wires]

    def receive(self, b):
        """receive (iknp.go:468-511) -> (bytes sent: the chunks' u-columns, concatenated; the labels)"""
        n = len(b)
        bbuf = bytearray((n + 7) // 8)
        for i, f in enumerate(b):
            if f:
                bbuf[i // 8] |= 1 << (i % 8)
        sent, result, ofs = b"", [], 0
        while ofs < n:
            rows = min(CHUNK_ROWS, n - ofs)
            br = (rows + 7) // 8
            chunk, out = b"", b""
            for i in range(K):
                t0 = self.g0[i].bytes(br)
                tmp = self.g1[i].bytes(br)
                tmp = _xor_prefix(tmp, t0)
                tmp = _xor_prefix(tmp, bytes(bbuf[ofs // 8:]))
                chunk += t0
                out += tmp
            sent += out
            result += create_labels(n - ofs, chunk, br)
            ofs += rows
        return sent, result

    def receive_bits(self, choices, n):
        """ReceiveBits (iknp.go:554-620): choices / result as lists of 64-bit words, bit i = word i // 64, bit i % 64.
        (As in the Go text: only the WHOLE 64-bit words of a chunk's byte rows take the choice bits — `words := byteRows / 8`.)"""
        res = [0] * ((n + 63) // 64)
        sent, ofs = b"", 0
        while ofs < n:
            rows = min(CHUNK_ROWS, n - ofs)
            br = (rows + 7) // 8
            word_off, words = ofs // 64, br // 8
            chunk, ucol = b"", b""
            for i in range(K):
                t0 = self.g0[i].bytes(br)
                tmp = bytearray(_xor_prefix(self.g1[i].bytes(br), t0))
                for w in range(words):
                    v = int.from_bytes(tmp[8 * w:8 * w + 8], "little") ^ choices[word_off + w]
                    tmp[8 * w:8 * w + 8] = v.to_bytes(8, "little")
                chunk += t0
                ucol += bytes(tmp)
            sent += ucol
            labels = create_labels(CHUNK_ROWS, chunk, br)
            for row in range(rows):
                if bit(labels[row], 0):
                    res[(ofs + row) // 64] |= 1 << ((ofs + row) % 64)
            ofs += rows
        return sent, res


class Sender:
    """IKNPSender (iknp.go:61-124): one PRG per base-OT label; k0[i] is the label the base OT delivered for bit i of delta"""

    def __init__(self, delta, k0):
        assert len(k0) == K
        self.delta = delta
        self.g0 = [Prg(k) for k in k0]

    def _columns(self, chunk, br):
        t = b""
        for i in range(K):
            col = self.g0[i].bytes(br)
            if bit(self.delta, i):
                col = _xor_prefix(col, chunk[i * br:])
            t += col
        return t

    def send(self, received, n):
        """send (iknp.go:197-226): `received` = the receiver's chunks, concatenated (each chunk is K * byteRows bytes: the
        receiver's rows per chunk are what this side can compute too)"""
        result, ofs, pos = [], 0, 0
        while ofs < n:
            br = (min(CHUNK_ROWS, n - ofs) + 7) // 8
            chunk = received[pos:pos + K * br]
            pos += K * br
            result += create_labels(n - ofs, self._columns(chunk, br), br)
            ofs += br * 8
        return result[:n]

    def send_bits(self, received, n):
        """SendBits (iknp.go:259-310): bit i of the result = bit (row % 8) of byte (row // 8) of column 0"""
        res = [0] * ((n + 63) // 64)
        ofs, pos = 0, 0
        while ofs < n:
            br = (min(CHUNK_ROWS, n - ofs) + 7) // 8
            chunk = received[pos:pos + K * br]
            pos += K * br
            col0 = self._columns(chunk, br)[:br]
            rows = min(br * 8, n - ofs)
            for row in range(rows):
                if (col0[row // 8] >> (row % 8)) & 1:
                    res[(ofs + row) // 64] |= 1 << ((ofs + row) % 64)
            ofs += rows
        return res


class Mitccrh:
    """MITCCRH (mitccrh.go:47-128): keys start ^ (gid, 0), gid counting on; key i of a batch hashes h consecutive blocks"""

    def __init__(self, seed, batch=OT_BATCH):
        self.batch, self.start, self.gid = batch, seed, 0
        self.ciphers, self.used = [None] * batch, batch

    def _renew(self):
        for i in range(self.batch):
            key = lxor((self.gid, 0), self.start)
            self.gid = (self.gid + 1) & M64
            self.ciphers[i] = AES(label_bytes(key))
        self.used = 0

    def hash(self, blks, k, h):
        assert k <= self.batch and self.batch % k == 0 and k * h == len(blks)
        if self.used == self.batch:
            self._renew()
        enc = [None] * len(blks)
        for i in range(k):
            c = self.ciphers[self.used + i]
            for j in range(h):
                enc[i * h + j] = label_from_bytes(c.encrypt(label_bytes(blks[i * h + j])))
        self.used += k
        return [lxor(b, e) for b, e in zip(blks, enc)]


def cot_send_pads(seed, delta, data, wires):
    """COT.Send's pad loop (cot.go:155-182): the 2n labels that go out, in order.  A batch is always hashed as 8 x 2 blocks:
    past the end the pad array still holds the batch before's ciphertexts (it is allocated once, cot.go:155)."""
    m = Mitccrh(seed)
    pad = [(0, 0)] * (2 * OT_BATCH)
    out = []
    for i in range(0, len(wires), OT_BATCH):
        end = min(i + OT_BATCH, len(wires))
        for j in range(i, end):
            pad[2 * (j - i)] = data[j]
            pad[2 * (j - i) + 1] = lxor(data[j], delta)
        pad = m.hash(pad, OT_BATCH, 2)
        for j in range(i, end):
            pad[2 * (j - i)] = lxor(pad[2 * (j - i)], wires[j][0])
            pad[2 * (j - i) + 1] = lxor(pad[2 * (j - i) + 1], wires[j][1])
        out += pad[:2 * (end - i)]
    return out


def cot_receive_unpad(seed, flags, sent, result):
    """COT.Receive's loop (cot.go:200-232): `sent` = the 2n labels of cot_send_pads; returns the n chosen labels.
    copy(pad, result[i:]) moves at most 8 labels: a short last batch keeps the batch before's tail in the pad array."""
    m = Mitccrh(seed)
    pad = [(0, 0)] * OT_BATCH
    out = list(result)
    for i in range(0, len(flags), OT_BATCH):
        end = min(OT_BATCH, len(flags) - i)
        src = result[i:i + OT_BATCH]
        pad = list(src) + pad[len(src):]
        pad = m.hash(pad, OT_BATCH, 1)
        for j in range(end):
            res0, res1 = sent[2 * (i + j)], sent[2 * (i + j) + 1]
            out[i + j] = lxor(res1 if flags[i + j] else res0, pad[j])
    return out


def rot_send(seed, delta, data):
    """ROT.Send's loop (rot.go:156-172): wires[j] = (H(data_j), H(data_j ^ delta))"""
    m = Mitccrh(seed)
    pad = [(0, 0)] * (2 * OT_BATCH)
    wires = []
    for i in range(0, len(data), OT_BATCH):
        end = min(i + OT_BATCH, len(data))
        for j in range(i, end):
            pad[2 * (j - i)] = data[j]
            pad[2 * (j - i) + 1] = lxor(data[j], delta)
        pad = m.hash(pad, OT_BATCH, 2)
        wires += [(pad[2 * (j - i)], pad[2 * (j - i) + 1]) for j in range(i, end)]
    return wires


def rot_receive(seed, result):
    """ROT.Receive's loop (rot.go:194-199)"""
    m = Mitccrh(seed)
    pad = [(0, 0)] * OT_BATCH
    out = list(result)
    for i in range(0, len(result), OT_BATCH):
        src = result[i:i + OT_BATCH]
        pad = list(src) + pad[len(src):]
        pad = m.hash(pad, OT_BATCH, 1)
        out[i:i + len(src)] = pad[:len(src)]
    return out


def clmul64(a, b):           # mul128_generic.go:28-42
    lo = hi = 0
    for i in range(64):
        if (b >> i) & 1:
            lo ^= (a << i) & M64
            if i:
                hi ^= a >> (64 - i)
    return lo, hi


def mul128(a, b):            # mul128Generic (mul128_generic.go:7-26): the 256-bit carry-less product, unreduced
    p00, p01, p10, p11 = clmul64(a[0], b[0]), clmul64(a[0], b[1]), clmul64(a[1], b[0]), clmul64(a[1], b[1])
    mid_lo, mid_hi = p01[0] ^ p10[0], p01[1] ^ p10[1]
    return (p00[0], p00[1] ^ mid_lo), (mid_hi ^ p11[0], p11[1])


def inner_product(a, b):     # vectorInnPrdtSumNoRed (gf128.go:9-22)
    r1 = r2 = (0, 0)
    for x, y in zip(a, b):
        lo, hi = mul128(x, y)
        r1, r2 = lxor(r1, lo), lxor(r2, hi)
    return r1, r2


def _chi_sums(seed2, result, choice_vector):
    """the chi-PRG walk both sides do (iknp.go:150-172 / 425-457): blocks of 1 024 labels over the result, then 256 more"""
    prg = Prg(seed2)
    q0 = q1 = (0, 0)
    chis = []
    for i in range(0, len(result), 1024):
        chi = prg.labels(min(1024, len(result) - i))
        r0, r1 = inner_product(chi, result[i:])
        q0, q1 = lxor(q0, r0), lxor(q1, r1)
        chis += chi
    chi = prg.labels(len(choice_vector))
    r0, r1 = inner_product(chi, choice_vector)
    return lxor(q0, r0), lxor(q1, r1), chis, chi


def kos_receiver_tags(seed2, result, b, choice_vector, bcv):
    """the receiver's half of the malicious check (iknp.go:405-465) -> (x, t0, t1)"""
    t0, t1, chis, chi_cv = _chi_sums(seed2, result, choice_vector)
    x = (0, 0)
    for c, f in zip(chis, b):
        if f:
            x = lxor(x, c)
    for c, f in zip(chi_cv, bcv):
        if f:
            x = lxor(x, c)
    return x, t0, t1


def kos_sender_check(seed2, result, choice_vector, delta, x, t0, t1):
    """the sender's half (iknp.go:138-194): q ^ x * delta == t"""
    q0, q1, _, _ = _chi_sums(seed2, result, choice_vector)
    r0, r1 = mul128(x, delta)
    return lxor(q0, r0) == t0 and lxor(q1, r1) == t1
