"""The reference's TestDeterministicTranscript and TestSessionIdempotency (sha2pc/sha2pc_test.go:73-130, :292-438) re-run
without Go — TEST INFRASTRUCTURE.

Those tests are the only place where the reference pins garbled-table BYTES: they run the four sha2pc rounds with three
deterministic byte streams and compare the SHA-256 of every encoded round with constants (`expRound1..3`, `expFinal`,
:120-124; `idemRound1..3Hash`, :413-416).  Round 3 carries the garbling key, all 42 914 table labels of sha256xor.mpclc,
the garbler's input labels, the output wires and the OT ciphertexts, so whoever reproduces `expRound3` garbles that
circuit byte for byte as the Go code does.  Everything the tests need besides `Circuit.Garble` is restated here from the
reference and from the published algorithms of the Go standard library it calls (go.mod: go 1.25.0; the library's source is
not under /root/reference):

  * math/rand (v1) `rand.New(rand.NewSource(seed))`: the additive lagged Fibonacci source x[n] = x[n-607] + x[n-273]
    mod 2^64, seeded through the Lehmer generator 48271 * x mod (2^31 - 1) and XORed with the 607-word table `rngCooked`
    ("the state of the generator after 780e10 iterations").  The table is not copied from anywhere: it is recomputed by
    tests/golden/gen_go_mathrand_cooked.py (the 7.8e12 steps as one power of t modulo t^607 - t^334 - 1 over Z/2^64: under
    a second; gen_go_mathrand_cooked.c is the same run step by step, 25 minutes) and committed as
    tests/golden/go_mathrand_cooked.txt.  `deterministicReader.Read` draws every byte as `Intn(256)` (:487-493) = bits
    32..39 of the source's next value.
  * crypto/rand.Int(reader, N): k = 32 bytes per try, big-endian, accepted when < N.
  * crypto/elliptic P-256 (ScalarBaseMult / ScalarMult / Add on affine big integers): plain Jacobian arithmetic below.
  * ot/co_helpers.go (GenerateCOSenderSetup :77, BuildCOChoices :143, EncryptCOCiphertexts :108, deriveMask :222),
    sha2pc/garbler.go:37-136, evaluator.go:27-60, encoding.go (EncodeRound1 :35, EncodeRound2 :84, EncodeRound3 :149 and the
    helpers they call), bits.go.
The hashes of rounds 1 and 2 (no garbling involved) pin all of that; round 3 then pins the garbling.

`garble` is a parameter: the tests pass the CPU oracle (tests/test_go_transcript.py) and the HIP engine
(tests/test_gpu_go_transcript.py); both must land on the Go constants.
"""
import hashlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# The two deterministic runs whose round hashes the reference's tests hold: the seed names of the three readers (garbler
# round 1, evaluator, garbler round 3) and the expected SHA-256 of the encoded rounds 1 - 3.
CASES = {
    # TestDeterministicTranscript, sha2pc/sha2pc_test.go:75-77 and :120-124
    "transcript": ((b"garbler-round1", b"evaluator-seed", b"garbler-round3"),
                   ("0191a7115a2ae1a1ff5ef7c9dbc5cf1078049b9e8fb77270b6b3c8f033220174",
                    "ff6286651743fff6b5b98857425fd11b9b2f877bb54258230054fdbe16575c84",
                    "ae10edf7fdb70a039b817cbacd9acf5069e3b019cf0d33eb48754536b2a7af39")),
    # TestSessionIdempotency, sha2pc/sha2pc_test.go:295-300 and :413-416
    "idempotency": ((b"garbler-r1-idem", b"eval-idem", b"garbler-r3-idem"),
                    ("8af8af7e7ea89c35fec609e54e519c956dc6c1775b3af16f534b812a5a40f3d9",
                     "19808c4e94f4103544b2bc4438e786ecebe0729e723d46baf8b3872d2ba36de8",
                     "80b05a7f8312d2247801498411217fcf21d5bd10767615979636770f057fa7cd")),
}
EXP_FINAL = "4b2f74579fc7c778745121996f604371a326dc5174f9851706032626668abf2e"  # :124, :418
ROUND3_LEN = 707146  # sha2pc_test.go:239

# ---- math/rand (v1) ------------------------------------------------------------------------------------------------------

_LEN, _TAP = 607, 273
_M31 = (1 << 31) - 1
_MASK64 = (1 << 64) - 1


def cooked_table():
    with open(os.path.join(HERE, "golden", "go_mathrand_cooked.txt")) as f:
        vals = [int(line) for line in f if line.strip() and not line.startswith("#")]
    assert len(vals) == _LEN
    return [v & _MASK64 for v in vals]


def _seedrand(x):
    hi, lo = divmod(x, 44488)
    x = 48271 * lo - 3399 * hi
    return x + _M31 if x < 0 else x


class GoRandSource:
    """rngSource of math/rand: Seed and Uint64"""

    def __init__(self, seed, cooked=None):
        cooked = cooked or cooked_table()
        self.tap, self.feed = 0, _LEN - _TAP
        seed %= _M31  # (Go's remainder has the dividend's sign and is then lifted by 2^31 - 1: the same value)
        if seed == 0:
            seed = 89482311
        x = seed
        self.vec = [0] * _LEN
        for i in range(-20, _LEN):
            x = _seedrand(x)
            if i >= 0:
                u = x << 40
                x = _seedrand(x)
                u ^= x << 20
                x = _seedrand(x)
                u ^= x
                self.vec[i] = (u ^ cooked[i]) & _MASK64

    def uint64(self):
        self.tap = self.tap - 1 if self.tap else _LEN - 1
        self.feed = self.feed - 1 if self.feed else _LEN - 1
        x = (self.vec[self.feed] + self.vec[self.tap]) & _MASK64
        self.vec[self.feed] = x
        return x


class DeterministicReader:
    """newDeterministicReader (sha2pc_test.go:478-493): seed = first 8 bytes of SHA-256(name) as a big-endian int64; every
    byte is Intn(256) = Int31() & 255 = bits 32..39 of the source's next value"""

    def __init__(self, name, cooked=None):
        seed = int.from_bytes(hashlib.sha256(name).digest()[:8], "big", signed=True)
        self.src = GoRandSource(seed, cooked)

    def read(self, n):
        nxt = self.src.uint64
        return bytes((nxt() >> 32) & 0xFF for _ in range(n))


# ---- P-256 -----------------------------------------------------------------------------------------------------------------

P = 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF
N = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551
B = 0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B
G = (0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
     0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5)


def on_curve(pt):
    x, y = pt
    return (y * y - (x * x * x - 3 * x + B)) % P == 0


def _dbl(p):
    x, y, z = p
    if not y or not z:
        return (0, 1, 0)
    zz = z * z % P
    m = 3 * (x - zz) * (x + zz) % P
    yy = y * y % P
    s = 4 * x * yy % P
    x3 = (m * m - 2 * s) % P
    return (x3, (m * (s - x3) - 8 * yy * yy) % P, 2 * y * z % P)


def _add(p, q):
    if not p[2]:
        return q
    if not q[2]:
        return p
    x1, y1, z1 = p
    x2, y2, z2 = q
    z1z1, z2z2 = z1 * z1 % P, z2 * z2 % P
    u1, u2 = x1 * z2z2 % P, x2 * z1z1 % P
    s1, s2 = y1 * z2 * z2z2 % P, y2 * z1 * z1z1 % P
    if u1 == u2:
        return _dbl(p) if s1 == s2 else (0, 1, 0)
    h, r = (u2 - u1) % P, (s2 - s1) % P
    hh = h * h % P
    hhh = h * hh % P
    v = u1 * hh % P
    x3 = (r * r - hhh - 2 * v) % P
    return (x3, (r * (v - x3) - s1 * hhh) % P, h * z1 * z2 % P)


def _affine(p):
    if not p[2]:
        return (0, 0)  # (crypto/elliptic's encoding of the point at infinity; not met by the transcript)
    zi = pow(p[2], -1, P)
    zi2 = zi * zi % P
    return (p[0] * zi2 % P, p[1] * zi2 * zi % P)


def scalar_mult(pt, k):
    """curve.ScalarMult(x, y, k.Bytes()): k is reduced mod N by the curve code"""
    k %= N
    acc, q = (0, 1, 0), (pt[0], pt[1], 1)
    while k:
        if k & 1:
            acc = _add(acc, q)
        q = _dbl(q)
        k >>= 1
    return _affine(acc)


def point_add(p, q):
    return _affine(_add((p[0], p[1], 1), (q[0], q[1], 1)))


def crand_int(reader, mx):
    """crypto/rand.Int: uniform in [0, mx) from `reader` by rejection"""
    bitlen = (mx - 1).bit_length()
    k, b = (bitlen + 7) // 8, bitlen % 8 or 8
    while True:
        raw = bytearray(reader.read(k))
        raw[0] &= (1 << b) - 1
        n = int.from_bytes(raw, "big")
        if n < mx:
            return n


def _bytes(v):  # big.Int.Bytes(): big-endian, no leading zeros
    return v.to_bytes((v.bit_length() + 7) // 8, "big")


def derive_mask(pt, idx):  # ot/co_helpers.go:222-236
    return hashlib.sha256(_bytes(pt[0]) + _bytes(pt[1]) + idx.to_bytes(8, "big")).digest()


# ---- the rounds ---------------------------------------------------------------------------------------------------------------

CURVE_CHUNK = b"\x05P-256"  # writeChunk([]byte(name)): uvarint length + bytes (encoding.go:663-668)


def bits_little(data):  # bits.go:4-15
    return [(byte >> k) & 1 for byte in data for k in range(8)]


def label_bytes(labels):
    """ot.Label.GetData of an array of {d0, d1}: BE(D0) || BE(D1) per label (label.go:105-108)"""
    a = np.ascontiguousarray(labels)
    out = np.empty((a.size, 2), ">u8")
    out[:, 0] = a["d0"].ravel()
    out[:, 1] = a["d1"].ravel()
    return out.tobytes()


def garbler_round1(reader):
    """GarblerRound1 (garbler.go:37-76) -> (encoded round 1, session)"""
    a = crand_int(reader, N)  # GenerateCOSenderSetup (co_helpers.go:77-105)
    A = scalar_mult(G, a)
    Aa = scalar_mult(A, a)
    sid = reader.read(8)
    enc = b"R1" + sid + CURVE_CHUNK + A[0].to_bytes(32, "big") + A[1].to_bytes(32, "big")
    return enc, {"sid": sid, "a": a, "A": A, "AaInv": (Aa[0], (P - Aa[1]) % P)}


def evaluator_round2(reader, sid, A, b_bytes):
    """EvaluatorRound2 (evaluator.go:27-60) + BuildCOChoices (co_helpers.go:143-181) -> (encoded round 2, points, scalars)"""
    points, scalars = [], []
    for bit in bits_little(b_bytes):
        s = crand_int(reader, N)
        pt = scalar_mult(G, s)
        if bit:
            pt = point_add(pt, A)
        points.append(pt)
        scalars.append(s)
    signs = bytearray(32)
    for i, pt in enumerate(points):  # packPointSigns (encoding.go:546-558)
        if pt[1] & 1:
            signs[i // 8] |= 1 << (i % 8)
    enc = b"R2" + sid + CURVE_CHUNK + b"".join(pt[0].to_bytes(32, "big") for pt in points) + bytes(signs)
    return enc, points, scalars


def garbler_round3(reader, session, a_bytes, points, circ, garble):
    """GarblerRound3 (garbler.go:80-136) + EncodeRound3 (encoding.go:149-174).
    garble(key, rnd) -> (wires as an array of {l0, l1} labels for at least the 512 inputs and the 256 outputs — a dict
    {"in": [512], "out": [256]} —, slab of all table labels in gate order)"""
    key = reader.read(32)
    nin = sum(circ.Inputs)
    rnd = reader.read(16 * (1 + nin))  # Circuit.Garble: R, then L0 of every input wire (garble.go:253-278)
    wires, slab = garble(key, rnd)
    win, wout = wires["in"], wires["out"]
    assert len(win) == 512 and len(wout) == 256 and len(slab) == 42914
    bits = bits_little(a_bytes)
    ginputs = b"".join(label_bytes(win[i]["l1" if bits[i] else "l0"]) for i in range(256))
    hints = b"".join(label_bytes(w["l0"]) + label_bytes(w["l1"]) for w in wout)
    cts = []
    for idx, pt in enumerate(points):  # EncryptCOCiphertexts (co_helpers.go:108-140)
        assert on_curve(pt)
        Bp = scalar_mult(pt, session["a"])
        Ba = point_add(Bp, session["AaInv"])
        w = win[256 + idx]
        m0, m1 = derive_mask(Bp, idx), derive_mask(Ba, idx)
        l0, l1 = label_bytes(w["l0"]), label_bytes(w["l1"])
        cts.append(bytes(x ^ y for x, y in zip(m0[:16], l0)) + bytes(x ^ y for x, y in zip(m1[:16], l1)))
    enc = b"R3" + session["sid"] + key + label_bytes(slab) + ginputs + hints + b"".join(cts)
    return enc, key, cts


def transcript(circ, garble, case="transcript", cooked=None):
    """rounds 1 - 3 of one of CASES: the three SHA-256 digests (hex) and the encoded round 3"""
    cooked = cooked or cooked_table()
    (n1, n2, n3), _ = CASES[case]
    a = bytes(range(32))
    b = bytes(32 - i for i in range(32))
    r1, session = garbler_round1(DeterministicReader(n1, cooked))
    r2, points, scalars = evaluator_round2(DeterministicReader(n2, cooked), session["sid"], session["A"], b)
    r3, key, cts = garbler_round3(DeterministicReader(n3, cooked), session, a, points, circ, garble)
    h = lambda x: hashlib.sha256(x).hexdigest()
    return {"round1": h(r1), "round2": h(r2), "round3": h(r3), "round1_len": len(r1), "round2_len": len(r2),
            "round3_bytes": r3, "key": key, "ciphertexts": cts,
            "scalars": scalars, "A": session["A"]}


def evaluator_round4(t, evaluate):
    """EvaluatorRound4 (evaluator.go:64-113) on the encoded round 3 of transcript(): DecodeRound3, DecryptCOCiphertexts
    (co_helpers.go:194-219), Circuit.Eval, BitFromLabel against the output hints -> the digest (hex).
    evaluate(key, slab {d0, d1}[42914], inputs {d0, d1}[512]) -> output labels {d0, d1}[256]"""
    label = np.dtype([("d0", "<u8"), ("d1", "<u8")])

    def labels(raw):
        be = np.frombuffer(raw, ">u8").reshape(-1, 2)
        out = np.zeros(len(be), label)
        out["d0"], out["d1"] = be[:, 0], be[:, 1]
        return out

    r3 = t["round3_bytes"]
    assert r3[:2] == b"R3" and len(r3) == ROUND3_LEN
    key, off = r3[10:42], 42
    slab = labels(r3[off:off + 42914 * 16])
    off += 42914 * 16
    inputs = np.zeros(512, label)
    inputs[:256] = labels(r3[off:off + 4096])
    hints = labels(r3[off + 4096:off + 4096 + 8192]).reshape(256, 2)
    cts = r3[off + 4096 + 8192:]
    b_bits = bits_little(bytes(32 - i for i in range(32)))
    for idx in range(256):
        mask = derive_mask(scalar_mult(t["A"], t["scalars"][idx]), idx)
        ct = cts[32 * idx + 16:32 * idx + 32] if b_bits[idx] else cts[32 * idx:32 * idx + 16]
        inputs[256 + idx] = labels(bytes(x ^ y for x, y in zip(mask[:16], ct)))[0]
    out = evaluate(key, slab, inputs)
    bits = []
    for j in range(256):
        assert out[j] == hints[j, 0] or out[j] == hints[j, 1], "output label %d is neither of the wire's labels" % j
        bits.append(1 if out[j] == hints[j, 1] else 0)
    return bytes(sum(bits[8 * i + k] << k for k in range(8)) for i in range(32)).hex()
