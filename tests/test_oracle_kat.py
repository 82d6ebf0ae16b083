"""Pins the CPU oracle (oracle/) against every known-answer vector available for this path.

FIPS-197 / OpenSSL pin the AES that Go's crypto/aes provides (un-vendored dependency);
the MITCCRH, label and mul128 vectors are the reference's own (ot/mitccrh_test.go:23-30,
ot/label_test.go:40-92, ot/mul128_test.go:14-52).
"""
import ctypes as C
import ctypes.util
import os

import numpy as np
import pytest

import oracle
from tests.util import drbg

H = bytes.fromhex

FIPS197 = [  # Appendix C.1 / C.2 / C.3
    ("000102030405060708090a0b0c0d0e0f", "69c4e0d86a7b0430d8cdb78070b4c55a"),
    ("000102030405060708090a0b0c0d0e0f1011121314151617", "dda97ca4864cdfe06eaf70a0ec0d7191"),
    ("000102030405060708090a0b0c0d0e0f101112131415161718191a1b1c1d1e1f", "8ea2b7ca516745bfeafc49904b496089"),
]


@pytest.mark.parametrize("portable", [False, True])
def test_fips197_appendix_c(portable):
    pt = H("00112233445566778899aabbccddeeff")
    for key, ct in FIPS197:
        assert oracle.aes_encrypt(H(key), pt, portable=portable) == H(ct)


def test_fips197_appendix_b():
    assert oracle.aes_encrypt(H("2b7e151628aed2a6abf7158809cf4f3c"), H("3243f6a8885a308d313198a2e0370734")) == H(
        "3925841d02dc09fbdc118597196a0b32")


def test_aes_key_size_error():
    # aes.NewCipher rejects anything but 16/24/32 bytes (garble.go:260, eval.go:20)
    for n in (0, 15, 17, 31, 33):
        with pytest.raises(oracle.OracleError) as e:
            oracle.aes_encrypt(bytes(n), bytes(16))
        assert e.value.code == oracle.E_KEYSIZE


def test_portable_matches_aesni():
    if not oracle.using_aesni():
        pytest.skip("no AES-NI on this host")
    for i in range(64):
        key = drbg("k%d" % i, (16, 24, 32)[i % 3])
        blk = drbg("b%d" % i, 16)
        assert oracle.aes_encrypt(key, blk, portable=True) == oracle.aes_encrypt(key, blk, portable=False)


def _openssl():
    name = ctypes.util.find_library("crypto")
    if not name:
        return None
    try:
        lib = C.CDLL(name)
        lib.EVP_CIPHER_CTX_new.restype = C.c_void_p
        for f in ("EVP_aes_128_ecb", "EVP_aes_192_ecb", "EVP_aes_256_ecb", "EVP_aes_128_ctr"):
            getattr(lib, f).restype = C.c_void_p
        return lib
    except OSError:
        return None


def _ossl_encrypt(lib, cipher, key, iv, data):
    ctx = C.c_void_p(lib.EVP_CIPHER_CTX_new())
    assert lib.EVP_EncryptInit_ex(ctx, C.c_void_p(getattr(lib, cipher)()), None, key, iv) == 1
    lib.EVP_CIPHER_CTX_set_padding(ctx, 0)
    out = C.create_string_buffer(len(data) + 32)
    n = C.c_int(0)
    assert lib.EVP_EncryptUpdate(ctx, out, C.byref(n), data, len(data)) == 1
    lib.EVP_CIPHER_CTX_free(ctx)
    return out.raw[: n.value]


def test_against_openssl():
    lib = _openssl()
    if lib is None:
        pytest.skip("libcrypto not loadable")
    for i in range(48):
        klen = (16, 24, 32)[i % 3]
        key, blk = drbg("ok%d" % i, klen), drbg("ob%d" % i, 16)
        want = _ossl_encrypt(lib, "EVP_aes_%d_ecb" % (8 * klen), key, None, blk)
        assert oracle.aes_encrypt(key, blk) == want
    # newPrg/prg (iknp.go:622-637): AES-128-CTR, zero IV, stream persists across calls,
    # including draws that are not a multiple of 16 bytes
    for i in range(4):
        key = drbg("ck%d" % i, 16)
        prg = oracle.Prg(oracle.label_from_bytes(key))
        got = b"".join(prg.bytes(n) for n in (64, 3, 13, 16, 1, 64, 24, 7))
        want = _ossl_encrypt(lib, "EVP_aes_128_ctr", key, bytes(16), bytes(len(got)))
        assert got == want


def test_ctr_prg_is_aes_of_counter():
    key = drbg("ctr", 16)
    prg = oracle.Prg(oracle.label_from_bytes(key))
    stream = prg.bytes(16 * 5)
    for j in range(5):
        assert stream[16 * j : 16 * j + 16] == oracle.aes_encrypt(key, j.to_bytes(16, "big"))
    labs = oracle.Prg(oracle.label_from_bytes(key)).labels(3)
    for j in range(3):
        assert oracle.label_to_bytes(labs[j]) == stream[16 * j : 16 * j + 16]


# ---- reference vectors -----------------------------------------------------------

MITCCRH_BLOCKS = [  # ot/mitccrh_test.go:23-30 (seed 0, batchSize 8, k=8, h=2, zero blocks)
    "66e94bd4ef8a2c3b884cfa59ca342b2e", "f6b7bdd1caeebab574683893c4475484",
    "5c76002bc7206560efe550c80b8f12cc", "ec331f5dd1c5f40e28ea541caec913f6",
    "932c6dbf69255cf13edcdb72233acea3", "6d5c3e022e5a6f7be663b9e69bcea443",
    "e013d7f4fa7abd93a7b85db9cfff9b14", "f0a2a65d245dd6199dc70951c2478b65",
]


def test_mitccrh_reference_vectors():
    m = oracle.MITCCRH((0, 0), 8)
    blks = np.zeros(16, oracle.LABEL)
    m.hash(blks, 8, 2)
    for i in range(8):
        for j in range(2):
            assert oracle.label_to_bytes(blks[i * 2 + j]).hex() == MITCCRH_BLOCKS[i]
    # the key of OT g is BE(Label{D0:g, D1:0} ^ seed)  (mitccrh.go:70-89)
    for g in range(8):
        key = g.to_bytes(8, "big") + bytes(8)
        assert oracle.aes_encrypt(key, bytes(16)).hex() == MITCCRH_BLOCKS[g]


def test_mitccrh_renews_keys_per_batch():
    seed = oracle.label_from_bytes(drbg("mseed", 16))
    m = oracle.MITCCRH(seed, 8)
    blks = np.zeros(8, oracle.LABEL)
    for batch in range(3):
        raw = [drbg("mb%d_%d" % (batch, i), 16) for i in range(8)]
        for i in range(8):
            blks[i] = oracle.label_from_bytes(raw[i])
        m.hash(blks, 8, 1)
        for i in range(8):
            gid = batch * 8 + i
            key = ((gid ^ seed[0]).to_bytes(8, "big")) + seed[1].to_bytes(8, "big")
            want = bytes(a ^ b for a, b in zip(raw[i], oracle.aes_encrypt(key, raw[i])))
            assert oracle.label_to_bytes(blks[i]) == want


def test_label_arithmetic_reference_values():
    F = 0xFFFFFFFFFFFFFFFF
    # ot/label_test.go:40-92
    assert oracle.label_set_s((F, F), True) == (F, F)
    assert oracle.label_set_s((F, F), False) == (0x7FFFFFFFFFFFFFFF, F)
    assert oracle.label_mul2((0, F)) == (0x1, 0xFFFFFFFFFFFFFFFE)
    assert oracle.label_mul4((0, F)) == (0x3, 0xFFFFFFFFFFFFFFFC)
    # Bit(i): i < 64 reads D0 (label.go:129-141)
    assert oracle.label_bit((1, 0), 0) == 1 and oracle.label_bit((0, 1), 64) == 1
    assert oracle.label_bit((1 << 63, 0), 63) == 1 and oracle.label_bit((0, 1 << 63), 127) == 1
    # GetData / SetData are big-endian D0 || D1 (label.go:105-114)
    b = H("0123456789abcdeffedcba9876543210")
    assert oracle.label_from_bytes(b) == (0x0123456789ABCDEF, 0xFEDCBA9876543210)
    assert oracle.label_to_bytes((0x0123456789ABCDEF, 0xFEDCBA9876543210)) == b


def test_encrypt_half_survey_cross_check():
    # SURVEY.md §8(a) self-check vector (survey-derived with OpenSSL, not from Go): catches endianness slips
    key = bytes(range(32))
    x = (0x0123456789ABCDEF, 0xFEDCBA9876543210)
    h = oracle.encrypt_half(key, x, 5)
    assert oracle.label_to_bytes(h).hex() == "fe48918e14f6a1c2bd2e75195d8956d3"
    # H(x,i) = AES(K) ^ K with K = 2x ^ i
    k = oracle.label_mul2(x)
    k = (k[0], k[1] ^ 5)
    kb = oracle.label_to_bytes(k)
    assert kb.hex() == "02468acf13579bdffdb97530eca86425"
    assert bytes(a ^ b for a, b in zip(oracle.aes_encrypt(key, kb), kb)) == oracle.label_to_bytes(h)


def test_encrypt_decrypt_roundtrip():
    # circuit/enc_test.go:19-43: decrypt(encrypt(c)) == c with a 32-byte zero key
    key = bytes(32)
    for t in range(16):
        a = oracle.label_from_bytes(drbg("a%d" % t, 16))
        b = oracle.label_from_bytes(drbg("b%d" % t, 16))
        c = oracle.label_from_bytes(drbg("c%d" % t, 16))
        e = oracle.encrypt(key, a, b, c, t)
        assert oracle.decrypt(key, a, b, t, e) == c


def test_mul128_reference_properties():
    # ot/mul128_test.go:14-52
    one, zero = (1, 0), (0, 0)
    x = oracle.label_from_bytes(drbg("mx", 16))
    assert oracle.mul128(x, one) == (x, zero)
    assert oracle.mul128(x, zero) == (zero, zero)
    lo, hi = oracle.mul128((1 << 63, 0), (1 << 63, 0))  # x^63 * x^63 = x^126
    assert lo == (0, 1 << 62) and hi == zero
    for i in range(50):
        a = oracle.label_from_bytes(drbg("ma%d" % i, 16))
        b = oracle.label_from_bytes(drbg("mb%d" % i, 16))
        assert oracle.mul128(a, b) == oracle.mul128(a, b, ref=True)
        assert oracle.mul128(a, b) == oracle.mul128(b, a)
