// aes_device.h — CDNA4 device-side AES (T-table in LDS) and 128-bit label helpers.
//
// A label is kept exactly as Go lays out ot.Label{D0,D1 uint64} in memory (little-endian
// limbs), loaded as one uint4:  x = D0 low32, y = D0 high32, z = D1 low32, w = D1 high32.
// The AES input block is BE(D0)||BE(D1) (ot/label.go:105-108), i.e. the four big-endian
// state columns are  s0 = y, s1 = x, s2 = w, s3 = z  — no byte swaps anywhere on the device.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace gc {

// LDS image of the round tables: Te0..Te3, 256 words each.
constexpr int kTeWords = 4 * 256;

__device__ __forceinline__ uint32_t rotr32(uint32_t v, uint32_t n) { return __builtin_amdgcn_alignbit(v, v, n); }

// Cooperative table load (blockDim.x must be a multiple of 256 or at least cover 256 entries
// through the stride loop).  g_te0 is the 1 KiB Te0 table in global memory (L2-resident).
__device__ __forceinline__ void load_te_tables(uint32_t *te, const uint32_t *__restrict__ g_te0) {
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        uint32_t t = g_te0[i];
        te[i] = t;
        te[256 + i] = rotr32(t, 8);
        te[512 + i] = rotr32(t, 16);
        te[768 + i] = rotr32(t, 24);
    }
}

// N independent blocks through NR rounds in lock-step (N-way ILP hides the LDS latency).
// s[k][0..3] are the big-endian columns; rk are big-endian round-key words (wave-uniform ->
// scalar loads).
template <int NR, int N>
__device__ __forceinline__ void aes_encrypt_n(uint32_t (&s)[N][4], const uint32_t *__restrict__ rk,
                                              const uint32_t *te) {
    const uint32_t *te0 = te, *te1 = te + 256, *te2 = te + 512, *te3 = te + 768;
#pragma unroll
    for (int k = 0; k < N; k++) {
        s[k][0] ^= rk[0];
        s[k][1] ^= rk[1];
        s[k][2] ^= rk[2];
        s[k][3] ^= rk[3];
    }
#pragma unroll
    for (int r = 1; r < NR; r++) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            uint32_t a0 = s[k][0], a1 = s[k][1], a2 = s[k][2], a3 = s[k][3];
            s[k][0] = te0[a0 >> 24] ^ te1[(a1 >> 16) & 0xff] ^ te2[(a2 >> 8) & 0xff] ^ te3[a3 & 0xff] ^ rk[4 * r + 0];
            s[k][1] = te0[a1 >> 24] ^ te1[(a2 >> 16) & 0xff] ^ te2[(a3 >> 8) & 0xff] ^ te3[a0 & 0xff] ^ rk[4 * r + 1];
            s[k][2] = te0[a2 >> 24] ^ te1[(a3 >> 16) & 0xff] ^ te2[(a0 >> 8) & 0xff] ^ te3[a1 & 0xff] ^ rk[4 * r + 2];
            s[k][3] = te0[a3 >> 24] ^ te1[(a0 >> 16) & 0xff] ^ te2[(a1 >> 8) & 0xff] ^ te3[a2 & 0xff] ^ rk[4 * r + 3];
        }
    }
    // final round: SubBytes + ShiftRows only; S[x] sits in Te2's top byte, Te3's byte 2,
    // Te0's byte 1 and Te1's low byte
#pragma unroll
    for (int k = 0; k < N; k++) {
        uint32_t a0 = s[k][0], a1 = s[k][1], a2 = s[k][2], a3 = s[k][3];
        s[k][0] = (te2[a0 >> 24] & 0xff000000u) ^ (te3[(a1 >> 16) & 0xff] & 0x00ff0000u) ^
                  (te0[(a2 >> 8) & 0xff] & 0x0000ff00u) ^ (te1[a3 & 0xff] & 0x000000ffu) ^ rk[4 * NR + 0];
        s[k][1] = (te2[a1 >> 24] & 0xff000000u) ^ (te3[(a2 >> 16) & 0xff] & 0x00ff0000u) ^
                  (te0[(a3 >> 8) & 0xff] & 0x0000ff00u) ^ (te1[a0 & 0xff] & 0x000000ffu) ^ rk[4 * NR + 1];
        s[k][2] = (te2[a2 >> 24] & 0xff000000u) ^ (te3[(a3 >> 16) & 0xff] & 0x00ff0000u) ^
                  (te0[(a0 >> 8) & 0xff] & 0x0000ff00u) ^ (te1[a1 & 0xff] & 0x000000ffu) ^ rk[4 * NR + 2];
        s[k][3] = (te2[a3 >> 24] & 0xff000000u) ^ (te3[(a0 >> 16) & 0xff] & 0x00ff0000u) ^
                  (te0[(a1 >> 8) & 0xff] & 0x0000ff00u) ^ (te1[a2 & 0xff] & 0x000000ffu) ^ rk[4 * NR + 3];
    }
}

// ---- bank-conflict-free variant ------------------------------------------------------------------
// One table (Te0) replicated 32x in LDS: entry x of copy c lives at word x*32 + c, so lane l always
// reads bank (l mod 32) — no two lanes of a 32-lane LDS group ever collide, whatever the indices
// (ds_read_b32: 2 LDS cycles per wave instruction instead of ~7 with random conflicts).
// Te1..Te3 are byte rotations of Te0 (one v_alignbit each).  32 KiB per workgroup.
constexpr int kTeReplWords = 256 * 32;

__device__ __forceinline__ void load_te_replicated(uint32_t *te, const uint32_t *__restrict__ g_te0) {
    // thread t writes copies (t & 31) of entries (t >> 5), (t >> 5) + blockDim/32, ...
    const uint32_t c = threadIdx.x & 31;
    for (uint32_t x = threadIdx.x >> 5; x < 256; x += blockDim.x >> 5) te[x * 32 + c] = g_te0[x];
}

// byte offset of this lane's copy inside an entry row
__device__ __forceinline__ uint32_t te_lane_off() { return (threadIdx.x & 31u) * 4u; }

__device__ __forceinline__ uint32_t te_at(const uint32_t *te, uint32_t x, uint32_t lane_off) {
    return *(const uint32_t *)((const char *)te + (x << 7) + lane_off);
}

template <int NR, int N>
__device__ __forceinline__ void aes_encrypt_repl(uint32_t (&s)[N][4], const uint32_t *__restrict__ rk,
                                                 const uint32_t *te, uint32_t lo) {
#pragma unroll
    for (int k = 0; k < N; k++) {
        s[k][0] ^= rk[0];
        s[k][1] ^= rk[1];
        s[k][2] ^= rk[2];
        s[k][3] ^= rk[3];
    }
#pragma unroll
    for (int r = 1; r < NR; r++) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            const uint32_t a0 = s[k][0], a1 = s[k][1], a2 = s[k][2], a3 = s[k][3];
#define GC_COL(b0, b1, b2, b3)                                                                               \
    (te_at(te, (b0) >> 24, lo) ^ rotr32(te_at(te, ((b1) >> 16) & 0xff, lo), 8) ^                          \
     rotr32(te_at(te, ((b2) >> 8) & 0xff, lo), 16) ^ rotr32(te_at(te, (b3)&0xff, lo), 24))
            s[k][0] = GC_COL(a0, a1, a2, a3) ^ rk[4 * r + 0];
            s[k][1] = GC_COL(a1, a2, a3, a0) ^ rk[4 * r + 1];
            s[k][2] = GC_COL(a2, a3, a0, a1) ^ rk[4 * r + 2];
            s[k][3] = GC_COL(a3, a0, a1, a2) ^ rk[4 * r + 3];
#undef GC_COL
        }
    }
    // final round: S[x] is byte 1 (and byte 2) of Te0[x]
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t a0 = s[k][0], a1 = s[k][1], a2 = s[k][2], a3 = s[k][3];
#define GC_SB(v) ((te_at(te, (v), lo) >> 8) & 0xffu)
#define GC_LAST(b0, b1, b2, b3)                                                                          \
    ((GC_SB((b0) >> 24) << 24) | (GC_SB(((b1) >> 16) & 0xff) << 16) | (GC_SB(((b2) >> 8) & 0xff) << 8) | \
     GC_SB((b3)&0xff))
        s[k][0] = GC_LAST(a0, a1, a2, a3) ^ rk[4 * NR + 0];
        s[k][1] = GC_LAST(a1, a2, a3, a0) ^ rk[4 * NR + 1];
        s[k][2] = GC_LAST(a2, a3, a0, a1) ^ rk[4 * NR + 2];
        s[k][3] = GC_LAST(a3, a0, a1, a2) ^ rk[4 * NR + 3];
#undef GC_LAST
#undef GC_SB
    }
}

// ---- perm-addressed dual table (the production AES of the fused kernels) ---------------------------
// LDS row x (256 bytes) = 32 copies of Te0[x] (bytes 0..127) followed by 32 copies of
// Te2[x] = rotr16(Te0[x]) (bytes 128..255); 64 KiB in all.  Lane l always reads copy (l mod 32), so
// bank = l mod 32 for every index: conflict-free by construction.  Because the index byte x sits
// byte-aligned at address bits [15:8], the whole LDS address is assembled by ONE v_perm_b32
// (state byte -> address byte 1, lane offset -> address byte 0) instead of a bfe + shift-or pair,
// and Te1/Te3 = rotr8(Te0/Te2) cost one v_alignbit per COLUMN:
//     col = Te0[b3(a0)] ^ Te2[b1(a2)] ^ rotr8(Te0[b2(a1)] ^ Te2[b0(a3)]) ^ rk
// -> 8 VALU (4 perm, xor3 with the key, xor, alignbit, xor) + 4 ds_read_b32 per column: 32 + 16 per round
// versus ~70 + 16 for the classic form.
constexpr int kTeDualBytes = 256 * 256;

__device__ __forceinline__ void te_dual_check(const uint32_t *te);

__device__ __forceinline__ void load_te_dual(uint32_t *te, const uint32_t *__restrict__ g_te0) {
    te_dual_check(te);
    const uint32_t c = threadIdx.x & 31;
    for (uint32_t x = threadIdx.x >> 5; x < 256; x += blockDim.x >> 5) {
        const uint32_t t = g_te0[x];
        te[x * 64 + c] = t;
        te[x * 64 + 32 + c] = rotr32(t, 16);
    }
}

// v_perm selectors: result byte1 = byte k of the state word (S0 -> selector 4+k), byte0 = lane offset
// (S1 byte 0), upper bytes zero (0x0c)
#define GC_PERM_SEL(k) (0x0c0c0000u | ((4u + (k)) << 8))

// The table MUST sit at LDS offset 0 (the kernels place it first in their only LDS allocation and call
// te_dual_check): the permuted word then IS the LDS byte address and the load needs no base add — with a
// pointer base the compiler spends one v_add per lookup, 16 of 52 VALU per round (tools/aes_model_ubench.hip:
// the round is VALU-bound at 4 waves/SIMD, 929 -> ~620 cycles per round per CU).
using lds_u32 = __attribute__((address_space(3))) const uint32_t;

__device__ __forceinline__ void te_dual_check(const uint32_t *te) {
    if ((uint32_t)(uintptr_t)te != 0u) __builtin_trap();  // low 32 bits of a flat LDS address = LDS offset
}

__device__ __forceinline__ uint32_t te_dual(uint32_t word, uint32_t sel, uint32_t lane_off) {
    const uint32_t addr = __builtin_amdgcn_perm(word, lane_off, sel);
    return *(lds_u32 *)(uintptr_t)addr;
}

// three-input XOR: one v_bitop3_b32 (gfx950)
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// One round's sixteen lookups of a block, issued as ONE batch: all addresses, then all loads, then the combines.
// Left to itself the compiler splits a round into two halves of eight lookups with a wait in between, i.e. two
// LDS round trips per round: 340 vs 249 cycles per round for a lone wave, 686 vs 663 with 16 waves per CU
// (tools/aes_model_ubench.hip, variants 1 / 8).  sched_barrier pins the three groups.
__device__ __forceinline__ void te_round_addrs(const uint32_t (&a)[4], uint32_t lo0, uint32_t lo2, uint32_t (&ad)[16]) {
    const uint32_t sel0 = GC_PERM_SEL(0), sel1 = GC_PERM_SEL(1), sel2 = GC_PERM_SEL(2), sel3 = GC_PERM_SEL(3);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        ad[4 * c + 0] = __builtin_amdgcn_perm(a[c], lo0, sel3);            // Te0[b3(a_c)]
        ad[4 * c + 1] = __builtin_amdgcn_perm(a[(c + 2) & 3], lo2, sel1);  // Te2[b1(a_c+2)]
        ad[4 * c + 2] = __builtin_amdgcn_perm(a[(c + 1) & 3], lo0, sel2);  // Te0[b2(a_c+1)]  (-> Te1 by rotr8)
        ad[4 * c + 3] = __builtin_amdgcn_perm(a[(c + 3) & 3], lo2, sel0);  // Te2[b0(a_c+3)]  (-> Te3 by rotr8)
    }
}

// rk: NR+1 round keys as big-endian words, expected in SGPRs (see load_round_keys)
// FEED = 1: the caller wants pi(K) ^ K (the fixed-key hash): the last AddRoundKey and the feed-forward of the input
// block become one three-input XOR per word.  FEED = 2: as 1, and the input is already whitened (s = K ^ rk[0..3],
// whiten_half / whiten_k) with the last round key folded (fold_last_round_key): pi(K) ^ K = SB ^ (rk_last ^ rk_0) ^ s.
template <int NR, int N, int FEED = 0>
__device__ __forceinline__ void aes_encrypt_dual(uint32_t (&s)[N][4], const uint32_t (&rk)[4 * (NR + 1)],
                                                 const uint32_t *te, uint32_t lo0) {
    uint32_t fin[FEED ? N : 1][4];
    if constexpr (FEED) {
#pragma unroll
        for (int k = 0; k < N; k++)
#pragma unroll
            for (int c = 0; c < 4; c++) fin[k][c] = s[k][c];
    }
    (void)te;  // the table sits at LDS offset 0 (te_dual_check)
    const uint32_t lo2 = lo0 + 128u;
    if constexpr (FEED != 2) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            s[k][0] ^= rk[0];
            s[k][1] ^= rk[1];
            s[k][2] ^= rk[2];
            s[k][3] ^= rk[3];
        }
    }
#pragma unroll
    for (int r = 1; r < NR; r++) {
        uint32_t ad[N][16], t[N][16];
#pragma unroll
        for (int k = 0; k < N; k++) te_round_addrs(s[k], lo0, lo2, ad[k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < N; k++)
#pragma unroll
            for (int i = 0; i < 16; i++) t[k][i] = *(lds_u32 *)(uintptr_t)ad[k][i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < N; k++)
#pragma unroll
            for (int c = 0; c < 4; c++)
                s[k][c] = xor3(t[k][4 * c], t[k][4 * c + 1], rk[4 * r + c]) ^ rotr32(t[k][4 * c + 2] ^ t[k][4 * c + 3], 8);
    }
    // final round (no MixColumns): S[x] is bytes 2,1 of Te0[x] and bytes 3,0 of Te2[x]; same byte positions as above
    // but the roles of the two table halves are swapped for the first and last byte
    {
        const uint32_t sel0 = GC_PERM_SEL(0), sel1 = GC_PERM_SEL(1), sel2 = GC_PERM_SEL(2), sel3 = GC_PERM_SEL(3);
        uint32_t ad[N][16], t[N][16];
#pragma unroll
        for (int k = 0; k < N; k++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                ad[k][4 * c + 0] = __builtin_amdgcn_perm(s[k][c], lo2, sel3);
                ad[k][4 * c + 1] = __builtin_amdgcn_perm(s[k][(c + 1) & 3], lo0, sel2);
                ad[k][4 * c + 2] = __builtin_amdgcn_perm(s[k][(c + 2) & 3], lo0, sel1);
                ad[k][4 * c + 3] = __builtin_amdgcn_perm(s[k][(c + 3) & 3], lo2, sel0);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < N; k++)
#pragma unroll
            for (int i = 0; i < 16; i++) t[k][i] = *(lds_u32 *)(uintptr_t)ad[k][i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < N; k++)
#pragma unroll
            for (int c = 0; c < 4; c++)
            {  // byte 3 of t0, byte 2 of t1, byte 1 of t2, byte 0 of t3: three bit-selects (v_bitop3 0xCA = m ? a : b)
                const uint32_t hi = __builtin_amdgcn_bitop3_b32(0xff000000u, t[k][4 * c], t[k][4 * c + 1], 0xCA);
                const uint32_t lo = __builtin_amdgcn_bitop3_b32(0x0000ff00u, t[k][4 * c + 2], t[k][4 * c + 3], 0xCA);
                const uint32_t sb = __builtin_amdgcn_bitop3_b32(0xffff0000u, hi, lo, 0xCA);
                if constexpr (FEED) s[k][c] = xor3(sb, rk[4 * NR + c], fin[k][c]);
                else s[k][c] = sb ^ rk[4 * NR + c];
            }
    }
}

// round keys -> scalar registers (the pointer is uniform; readfirstlane pins the values in SGPRs)
template <int NR>
__device__ __forceinline__ void load_round_keys(uint32_t (&rk)[4 * (NR + 1)], const uint32_t *__restrict__ g_rk) {
#pragma unroll
    for (int i = 0; i < 4 * (NR + 1); i++) rk[i] = __builtin_amdgcn_readfirstlane(g_rk[i]);
}

// As above, but only the first NS words go to SGPRs; the rest is fetched with vector loads (vz: a zero in a VGPR that
// the compiler cannot see through) and therefore lives in VGPRs.  The flat kernels have VGPRs to spare and are short
// of SGPRs: with all 60 AES-256 key words in SGPRs 25-40 other scalars spill, and every reload is a v_readlane — a
// VALU slot in kernels that are VALU-issue bound.
template <int NR, int NS>
__device__ __forceinline__ void load_round_keys_split(uint32_t (&rk)[4 * (NR + 1)], const uint32_t *__restrict__ g_rk,
                                                      uint32_t vz) {
#pragma unroll
    for (int i = 0; i < 4 * (NR + 1); i++)
        rk[i] = i < NS ? (uint32_t)__builtin_amdgcn_readfirstlane(g_rk[i]) : g_rk[i + vz];
}

// N hashes pi(K) ^ K in lock-step (N independent AES chains per lane hide the LDS latency)
template <int NR, int N>
__device__ __forceinline__ void hash_dual_n(const uint32_t (&k)[N][4], uint4 (&out)[N],
                                            const uint32_t (&rk)[4 * (NR + 1)], const uint32_t *te, uint32_t lo0) {
    uint32_t s[N][4];
#pragma unroll
    for (int i = 0; i < N; i++) {
        s[i][0] = k[i][0];
        s[i][1] = k[i][1];
        s[i][2] = k[i][2];
        s[i][3] = k[i][3];
    }
    aes_encrypt_dual<NR, N, 1>(s, rk, te, lo0);
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = make_uint4(s[i][1], s[i][0], s[i][3], s[i][2]);
}

template <int NR>
__device__ __forceinline__ uint4 hash_dual(const uint32_t (&k)[4], const uint32_t (&rk)[4 * (NR + 1)],
                                           const uint32_t *te, uint32_t lo0) {
    uint32_t s[1][4] = {{k[0], k[1], k[2], k[3]}};
    aes_encrypt_dual<NR, 1, 1>(s, rk, te, lo0);
    return make_uint4(s[0][1], s[0][0], s[0][3], s[0][2]);
}

// ---- label arithmetic (ot/label.go) on the uint4 form --------------------------------------

__device__ __forceinline__ uint4 lxor(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
__device__ __forceinline__ uint4 land(uint4 a, uint32_t m) { return make_uint4(a.x & m, a.y & m, a.z & m, a.w & m); }
// Label.S(): MSB of D0 (label.go:65-67)
__device__ __forceinline__ bool lbit_s(uint4 a) { return (a.y >> 31) != 0; }
// all-ones mask if S is set
__device__ __forceinline__ uint32_t smask(uint4 a) { return (uint32_t)((int32_t)a.y >> 31); }
__device__ __forceinline__ bool leq(uint4 a, uint4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

// K = (x << sh) as big-endian columns; sh = 1 (Mul2, label.go:79-83) or 2 (Mul4, :86-90)
template <int SH>
__device__ __forceinline__ void label_shl_cols(uint4 v, uint32_t (&k)[4]) {
    // columns: s0 = D0 hi (y), s1 = D0 lo (x), s2 = D1 hi (w), s3 = D1 lo (z)
    k[0] = __builtin_amdgcn_alignbit(v.y, v.x, 32 - SH);
    k[1] = __builtin_amdgcn_alignbit(v.x, v.w, 32 - SH);
    k[2] = __builtin_amdgcn_alignbit(v.w, v.z, 32 - SH);
    k[3] = v.z << SH;
}

__device__ __forceinline__ uint4 cols_to_label(const uint32_t (&s)[4]) { return make_uint4(s[1], s[0], s[3], s[2]); }

// ---- the same hash with the key whitening folded into the key set-up (flat kernels) --------
// rk_last ^= rk_0, once per kernel: the feed-forward then takes the whitened block instead of K
template <int NR>
__device__ __forceinline__ void fold_last_round_key(uint32_t (&rk)[4 * (NR + 1)]) {
#pragma unroll
    for (int c = 0; c < 4; c++) rk[4 * NR + c] ^= rk[c];
}
template <int NR>
__device__ __forceinline__ uint4 hash_dual_whitened(const uint32_t (&s0)[4], const uint32_t (&rk)[4 * (NR + 1)],
                                                    const uint32_t *te, uint32_t lo0) {
    uint32_t s[1][4] = {{s0[0], s0[1], s0[2], s0[3]}};
    aes_encrypt_dual<NR, 1, 2>(s, rk, te, lo0);
    return make_uint4(s[0][1], s[0][0], s[0][3], s[0][2]);
}

// ---- column-sliced form: the four lanes of a quad hold the four state columns of ONE block ----------------------
// For hash phases with so little work that the phase IS the latency of a lone wave's AES (deep, narrow circuits:
// sha256xor x 256 has ~13 ANDs per phase; a chain of ANDs has one): a lane looks up the four bytes of ITS column and
// the quad exchanges the table values with DPP quad permutes (fused into the XORs) — 4 look-ups + 10 VALU per round
// and lane instead of 16 + 32, four times the lanes.  Round keys come from LDS (keyaddr = byte address of word
// [round 0][column of this lane], 16 bytes per round; the last round key is folded, fold_last_round_key).
// s0 = the lane's column of K ^ rk_0; returns the lane's column of pi(K) ^ K.
constexpr int GC_QROT1 = 0x39;  // quad_perm [1,2,3,0]: value of the next column's lane
constexpr int GC_QROT2 = 0x4E;  // [2,3,0,1]
constexpr int GC_QROT3 = 0x93;  // [3,0,1,2]
template <int CTRL>
__device__ __forceinline__ uint32_t quad_from(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
template <int NR>
__device__ __forceinline__ uint32_t hash_col_whitened(uint32_t s0, uint32_t keyaddr, uint32_t lo0) {
    const uint32_t lo2 = lo0 + 128u;
    const uint32_t sel0 = GC_PERM_SEL(0), sel1 = GC_PERM_SEL(1), sel2 = GC_PERM_SEL(2), sel3 = GC_PERM_SEL(3);
    uint32_t s = s0;
#pragma unroll
    for (int r = 1; r < NR; r++) {
        // column c: Te0[b3(a_c)] ^ Te1[b2(a_c+1)] ^ Te2[b1(a_c+2)] ^ Te3[b0(a_c+3)], Te1/Te3 = rotr8(Te0/Te2)
        const uint32_t a0 = __builtin_amdgcn_perm(s, lo0, sel3), a1 = __builtin_amdgcn_perm(s, lo0, sel2);
        const uint32_t a2 = __builtin_amdgcn_perm(s, lo2, sel1), a3 = __builtin_amdgcn_perm(s, lo2, sel0);
        const uint32_t A = *(lds_u32 *)(uintptr_t)a0, B = *(lds_u32 *)(uintptr_t)a1;
        const uint32_t C = *(lds_u32 *)(uintptr_t)a2, D = *(lds_u32 *)(uintptr_t)a3;
        const uint32_t k = *(lds_u32 *)(uintptr_t)(keyaddr + 16u * r);
        const uint32_t x = (A ^ k) ^ quad_from<GC_QROT2>(C);
        const uint32_t y = quad_from<GC_QROT1>(B) ^ quad_from<GC_QROT3>(D);
        s = x ^ rotr32(y, 8);
    }
    // last round: byte 3 = S[b3(a_c)] (Te2), byte 2 = S[b2(a_c+1)] (Te0), byte 1 = S[b1(a_c+2)] (Te0), byte 0 (Te2)
    const uint32_t a0 = __builtin_amdgcn_perm(s, lo2, sel3), a1 = __builtin_amdgcn_perm(s, lo0, sel2);
    const uint32_t a2 = __builtin_amdgcn_perm(s, lo0, sel1), a3 = __builtin_amdgcn_perm(s, lo2, sel0);
    const uint32_t A = *(lds_u32 *)(uintptr_t)a0, B = *(lds_u32 *)(uintptr_t)a1;
    const uint32_t C = *(lds_u32 *)(uintptr_t)a2, D = *(lds_u32 *)(uintptr_t)a3;
    const uint32_t k = *(lds_u32 *)(uintptr_t)(keyaddr + 16u * NR);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32(0xff000000u, A, quad_from<GC_QROT1>(B), 0xCA);
    const uint32_t lo = __builtin_amdgcn_bitop3_b32(0x0000ff00u, quad_from<GC_QROT2>(C), quad_from<GC_QROT3>(D), 0xCA);
    return xor3(__builtin_amdgcn_bitop3_b32(0xffff0000u, hi, lo, 0xCA), k, s0);
}

// Hash inputs of encryptHalf (circuit/garble.go:104-136): K = 2x ^ i, i in the low 32 bits of D1
__device__ __forceinline__ void make_k_half(uint4 x, uint32_t tweak, uint32_t (&k)[4]) {
    label_shl_cols<1>(x, k);
    k[3] ^= tweak;
}
// makeK (circuit/garble.go:74-83): K = 2a ^ 4b ^ t
__device__ __forceinline__ void make_k(uint4 a, uint4 b, uint32_t tweak, uint32_t (&k)[4]) {
    uint32_t kb[4];
    label_shl_cols<1>(a, k);
    label_shl_cols<2>(b, kb);
    k[0] ^= kb[0];
    k[1] ^= kb[1];
    k[2] ^= kb[2];
    k[3] ^= kb[3] ^ tweak;
}

// whitened forms for hash_dual_whitened: s0 = K ^ rk_0, the tweak joins the last column's XOR
__device__ __forceinline__ void whiten_half(uint4 x, uint32_t tweak, const uint32_t *rk0, uint32_t (&s0)[4]) {
    uint32_t k[4];
    label_shl_cols<1>(x, k);
    s0[0] = k[0] ^ rk0[0];
    s0[1] = k[1] ^ rk0[1];
    s0[2] = k[2] ^ rk0[2];
    s0[3] = xor3(k[3], tweak, rk0[3]);
}
__device__ __forceinline__ void whiten_k(uint4 a, uint4 b, uint32_t tweak, const uint32_t *rk0, uint32_t (&s0)[4]) {
    uint32_t ka[4], kb[4];
    label_shl_cols<1>(a, ka);
    label_shl_cols<2>(b, kb);
    s0[0] = xor3(ka[0], kb[0], rk0[0]);
    s0[1] = xor3(ka[1], kb[1], rk0[1]);
    s0[2] = xor3(ka[2], kb[2], rk0[2]);
    s0[3] = xor3(ka[3], kb[3], rk0[3]) ^ tweak;
}

// N hashes pi(K) ^ K in lock-step; k[][] in: hash inputs, out: hash values (as labels)
template <int NR, int N>
__device__ __forceinline__ void hash_n(uint32_t (&k)[N][4], uint4 (&out)[N], const uint32_t *__restrict__ rk,
                                       const uint32_t *te) {
    uint32_t s[N][4];
#pragma unroll
    for (int i = 0; i < N; i++) {
        s[i][0] = k[i][0];
        s[i][1] = k[i][1];
        s[i][2] = k[i][2];
        s[i][3] = k[i][3];
    }
    aes_encrypt_n<NR, N>(s, rk, te);
#pragma unroll
    for (int i = 0; i < N; i++) {
        s[i][0] ^= k[i][0];
        s[i][1] ^= k[i][1];
        s[i][2] ^= k[i][2];
        s[i][3] ^= k[i][3];
        out[i] = cols_to_label(s[i]);
    }
}

// same as hash_n but through the conflict-free replicated table
template <int NR, int N>
__device__ __forceinline__ void hash_repl(uint32_t (&k)[N][4], uint4 (&out)[N], const uint32_t *__restrict__ rk,
                                          const uint32_t *te, uint32_t lo) {
    uint32_t s[N][4];
#pragma unroll
    for (int i = 0; i < N; i++) {
        s[i][0] = k[i][0];
        s[i][1] = k[i][1];
        s[i][2] = k[i][2];
        s[i][3] = k[i][3];
    }
    aes_encrypt_repl<NR, N>(s, rk, te, lo);
#pragma unroll
    for (int i = 0; i < N; i++) {
        s[i][0] ^= k[i][0];
        s[i][1] ^= k[i][1];
        s[i][2] ^= k[i][2];
        s[i][3] ^= k[i][3];
        out[i] = cols_to_label(s[i]);
    }
}

}  // namespace gc
