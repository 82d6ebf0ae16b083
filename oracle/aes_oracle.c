/*
 * aes_oracle.c — FIPS-197 AES-128/192/256 block encryption (oracle, test infrastructure).
 *
 * Restates what Go's crypto/aes (go 1.25.0, outside /root/reference) computes at the
 * reference call sites circuit/garble.go:260,122,46,63, circuit/eval.go:20,
 * ot/iknp.go:624-629 and ot/mitccrh.go:82,117: aes.NewCipher(key) with a 16/24/32-byte
 * key and Block.Encrypt of one 16-byte block.  Written from the FIPS-197 text:
 * the S-box is derived from GF(2^8) inversion + the affine map (§5.1.1), the key
 * schedule from §5.2, the cipher from §5.1.  An AES-NI path (same round keys) is used
 * for the CPU baseline timing; both paths are cross-checked in tests.
 */
#include "oracle.h"

#include <string.h>

#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#define ORC_HAVE_X86 1
#endif

static uint8_t SBOX[256];
static int sbox_ready;
static int force_portable;
static int aesni_ok = -1;

static uint8_t gf_mul(uint8_t a, uint8_t b) {
    uint8_t p = 0;
    for (int i = 0; i < 8; i++) {
        if (b & 1) p ^= a;
        uint8_t hi = a & 0x80;
        a = (uint8_t)(a << 1);
        if (hi) a ^= 0x1b; /* x^8 + x^4 + x^3 + x + 1 */
        b >>= 1;
    }
    return p;
}

static void sbox_init(void) {
    if (sbox_ready) return;
    for (int x = 0; x < 256; x++) {
        /* multiplicative inverse by exhaustive search (0 maps to 0) */
        uint8_t inv = 0;
        if (x) {
            for (int y = 1; y < 256; y++) {
                if (gf_mul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
            }
        }
        /* affine transformation, FIPS-197 eq. (5.1) */
        uint8_t s = 0;
        for (int i = 0; i < 8; i++) {
            int bit = ((inv >> i) ^ (inv >> ((i + 4) & 7)) ^ (inv >> ((i + 5) & 7)) ^
                       (inv >> ((i + 6) & 7)) ^ (inv >> ((i + 7) & 7)) ^ (0x63 >> i)) & 1;
            s |= (uint8_t)(bit << i);
        }
        SBOX[x] = s;
    }
    sbox_ready = 1;
}

void orc_aes_force_portable(int on) { force_portable = on; }

static int have_aesni(void) {
#ifdef ORC_HAVE_X86
    if (aesni_ok < 0) {
        unsigned a, b, c, d;
        aesni_ok = 0;
        if (__get_cpuid(1, &a, &b, &c, &d)) aesni_ok = (c >> 25) & 1;
    }
    return aesni_ok;
#else
    return 0;
#endif
}

int orc_aes_using_aesni(void) { return !force_portable && have_aesni(); }

int orc_aes_init(orc_aes *a, const uint8_t *key, size_t keylen) {
    if (keylen != 16 && keylen != 24 && keylen != 32) return ORC_E_KEYSIZE;
    sbox_init();
    int nk = (int)keylen / 4;
    int nr = nk + 6;
    a->rounds = nr;
    uint8_t *w = a->rk; /* w[i] = 4 bytes at w + 4*i */
    memcpy(w, key, keylen);
    uint8_t rcon = 1;
    for (int i = nk; i < 4 * (nr + 1); i++) {
        uint8_t t[4];
        memcpy(t, w + 4 * (i - 1), 4);
        if (i % nk == 0) {
            uint8_t r0 = t[0];
            t[0] = SBOX[t[1]] ^ rcon;
            t[1] = SBOX[t[2]];
            t[2] = SBOX[t[3]];
            t[3] = SBOX[r0];
            rcon = gf_mul(rcon, 2);
        } else if (nk > 6 && i % nk == 4) {
            for (int j = 0; j < 4; j++) t[j] = SBOX[t[j]];
        }
        for (int j = 0; j < 4; j++) w[4 * i + j] = w[4 * (i - nk) + j] ^ t[j];
    }
    return ORC_OK;
}

static void encrypt_portable(const orc_aes *a, const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ a->rk[i];
    for (int r = 1; r <= a->rounds; r++) {
        /* SubBytes + ShiftRows: state byte (row r, col c) is s[4c + r] */
        for (int c = 0; c < 4; c++)
            for (int row = 0; row < 4; row++) t[4 * c + row] = SBOX[s[4 * ((c + row) & 3) + row]];
        if (r != a->rounds) {
            for (int c = 0; c < 4; c++) {
                uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c + 0] = gf_mul(a0, 2) ^ gf_mul(a1, 3) ^ a2 ^ a3;
                s[4 * c + 1] = a0 ^ gf_mul(a1, 2) ^ gf_mul(a2, 3) ^ a3;
                s[4 * c + 2] = a0 ^ a1 ^ gf_mul(a2, 2) ^ gf_mul(a3, 3);
                s[4 * c + 3] = gf_mul(a0, 3) ^ a1 ^ a2 ^ gf_mul(a3, 2);
            }
        } else {
            memcpy(s, t, 16);
        }
        for (int i = 0; i < 16; i++) s[i] ^= a->rk[16 * r + i];
    }
    memcpy(out, s, 16);
}

#ifdef ORC_HAVE_X86
__attribute__((target("aes,sse2"))) static void encrypt_aesni(const orc_aes *a, const uint8_t in[16],
                                                              uint8_t out[16]) {
    const __m128i *rk = (const __m128i *)a->rk;
    __m128i s = _mm_xor_si128(_mm_loadu_si128((const __m128i *)in), _mm_loadu_si128(rk));
    for (int r = 1; r < a->rounds; r++) s = _mm_aesenc_si128(s, _mm_loadu_si128(rk + r));
    s = _mm_aesenclast_si128(s, _mm_loadu_si128(rk + a->rounds));
    _mm_storeu_si128((__m128i *)out, s);
}
#endif

void orc_aes_encrypt(const orc_aes *a, const uint8_t in[16], uint8_t out[16]) {
#ifdef ORC_HAVE_X86
    if (!force_portable && have_aesni()) {
        encrypt_aesni(a, in, out);
        return;
    }
#endif
    encrypt_portable(a, in, out);
}
