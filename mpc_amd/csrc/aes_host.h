// aes_host.h — host-side AES key schedule and T-table generation for the device kernels.
//
// The fixed-key cipher of the reference is Go's aes.NewCipher(key) (circuit/garble.go:260,
// circuit/eval.go:20, ot/iknp.go:624, ot/mitccrh.go:82).  The device keeps the state as four
// big-endian 32-bit columns, so round keys are produced as big-endian words w[0..4*(Nr+1)) and the
// round function uses the classic T-table Te0[x] = {02·S[x], S[x], S[x], 03·S[x]} (MSB first);
// Te1..Te3 are byte rotations of Te0.  Everything is generated from the field arithmetic at
// start-up (exp/log tables over generator 0x03) — no constant tables are embedded.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace gc {

struct AesTables {
    uint8_t sbox[256];
    uint32_t te0[256];
    AesTables() {
        uint8_t exp[256], log[256];
        uint8_t x = 1;
        for (int i = 0; i < 255; i++) {
            exp[i] = x;
            log[x] = (uint8_t)i;
            // multiply by the generator 0x03 = x * 2 ^ x
            uint8_t x2 = (uint8_t)((x << 1) ^ ((x & 0x80) ? 0x1b : 0));
            x = (uint8_t)(x2 ^ x);
        }
        exp[255] = exp[0];
        log[0] = 0;
        for (int v = 0; v < 256; v++) {
            uint8_t inv = v ? exp[(255 - log[v]) % 255] : 0;
            uint8_t s = inv;
            uint8_t r = inv;
            for (int k = 0; k < 4; k++) {
                r = (uint8_t)((r << 1) | (r >> 7));
                s ^= r;
            }
            s ^= 0x63;
            sbox[v] = s;
        }
        for (int v = 0; v < 256; v++) {
            uint32_t s = sbox[v];
            uint32_t s2 = ((s << 1) ^ ((s & 0x80) ? 0x11b : 0)) & 0xff;
            uint32_t s3 = s2 ^ s;
            te0[v] = (s2 << 24) | (s << 16) | (s << 8) | s3;
        }
    }
};

inline const AesTables &aes_tables() {
    static const AesTables t;
    return t;
}

// Expanded key: big-endian words, 4*(rounds+1) of them.
struct AesKey {
    uint32_t w[60];
    int rounds;  // 10 / 12 / 14, 0 = invalid key size
};

inline bool aes_expand_key(const uint8_t *key, size_t keylen, AesKey *out) {
    if (keylen != 16 && keylen != 24 && keylen != 32) {
        out->rounds = 0;
        return false;
    }
    const AesTables &T = aes_tables();
    const int nk = (int)keylen / 4, nr = nk + 6;
    out->rounds = nr;
    std::memset(out->w, 0, sizeof out->w);
    for (int i = 0; i < nk; i++)
        out->w[i] = ((uint32_t)key[4 * i] << 24) | ((uint32_t)key[4 * i + 1] << 16) | ((uint32_t)key[4 * i + 2] << 8) |
                    (uint32_t)key[4 * i + 3];
    auto subword = [&](uint32_t v) {
        return ((uint32_t)T.sbox[v >> 24] << 24) | ((uint32_t)T.sbox[(v >> 16) & 0xff] << 16) |
               ((uint32_t)T.sbox[(v >> 8) & 0xff] << 8) | (uint32_t)T.sbox[v & 0xff];
    };
    uint32_t rcon = 0x01000000u;
    for (int i = nk; i < 4 * (nr + 1); i++) {
        uint32_t t = out->w[i - 1];
        if (i % nk == 0) {
            t = subword((t << 8) | (t >> 24)) ^ rcon;
            uint32_t hi = rcon >> 24;
            hi = ((hi << 1) ^ ((hi & 0x80) ? 0x11b : 0)) & 0xff;
            rcon = hi << 24;
        } else if (nk > 6 && i % nk == 4) {
            t = subword(t);
        }
        out->w[i] = out->w[i - nk] ^ t;
    }
    return true;
}

}  // namespace gc
