// iknp_fused_kernels.hip — IKNP OT extension, one launch per call: column AES-128-CTR PRG + u-matrix / delta fold +
// the 128 x 512 bit transpose (createLabels), chunk by chunk, without a trip of the t-matrix through HBM.
//
//   receiver (ot/iknp.go:468-511):  t_i = PRG(g0_i), u_i = t_i ^ PRG(g1_i) ^ choice bytes -> u_out;  labels = transpose(t)
//   sender   (ot/iknp.go:197-226):  t_i = PRG(g0_i) ^ (delta.Bit(i) ? u_i : 0);                      labels = transpose(t)
//   PRG      (ot/iknp.go:622-645):  AES-128-CTR, key = BE(base-OT label), 128-bit big-endian counter, zero IV,
//                                   byte position persistent across chunks and calls
//   createLabels (ot/iknp.go:647-683): label 8*row+bit has bit j = bit `bit` of chunk[j*w + row]
//
// A 1024-thread workgroup walks groups of chunks (persistent grid).  Lane = (slice, column): every lane produces the
// 64 stream bytes of ITS column for one chunk — 4 AES blocks in lock-step (5 when the stream position is not a
// multiple of 16) with the column's round keys read from LDS (one ds_read_b128 per round for all blocks of the
// lane) and the perm-addressed dual T-table of the garbling kernels.  Sender: 8 slices = 8 chunks per step;
// receiver: 4 chunks x 2 streams (the g1 lanes hand their keystream to the g0 lanes through the chunk buffer).
// The chunk (128 columns x 16 dwords) is stored swizzled in LDS; one wave per chunk then transposes it with 64
// in-register 32 x 32 bit transposes (5 masked-swap stages — 160 VALU per 512 labels, where a ballot per label bit
// costs 2 x 128) and writes the labels back through the same 8 KiB of LDS so that the global store is coalesced.
// LDS: 64 KiB table | 22 KiB (44 KiB) round keys | 8 (4) chunk buffers of 8 448 bytes.
#include "aes_device.h"
#include "kernels.h"
#include <cstdlib>

namespace gc {

namespace {

constexpr int IKT = 1024;
constexpr uint32_t kKeyBytes = 128 * 176;  // [column][11 round keys][16 bytes]

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
using lds_u4 = __attribute__((address_space(3))) u32x4;
using lds_w32 = __attribute__((address_space(3))) uint32_t;

__device__ __forceinline__ uint4 lds_ld4(uint32_t addr) {
    const u32x4 v = *(lds_u4 *)(uintptr_t)addr;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_st4(uint32_t addr, uint4 v) {
    u32x4 w;
    w.x = v.x, w.y = v.y, w.z = v.z, w.w = v.w;
    *(lds_u4 *)(uintptr_t)addr = w;
}
__device__ __forceinline__ uint32_t lds_ld1(uint32_t addr) { return *(lds_w32 *)(uintptr_t)addr; }
__device__ __forceinline__ void lds_st1(uint32_t addr, uint32_t v) { *(lds_w32 *)(uintptr_t)addr = v; }

// Position (in dwords) of dword rd of column col inside a chunk buffer (kChunkBuf bytes).  Chosen so that
//   * the lane = column ds_write_b128 of a keystream quarter hits 8 distinct 16-byte bank slots per 8 lanes,
//   * the transposing wave (lane = rd * 4 + column group) reads dword rd of column (cg * 32 + k) from 32 distinct
//     banks per 32 lanes for every k, and
//   * for a fixed lane the 32 read addresses are one of FOUR bases plus the immediate 64 k (4 address registers
//     instead of 32).
constexpr uint32_t kCgStride = 520;              // dwords per column group: 32 columns x 16 + 8 (bank skew)
constexpr uint32_t kChunkBuf = 512 * 16 + 16 * 16;  // 8 448 bytes: >= 4 * kCgStride * 4 for the columns, and room for the
                                                   // 512 labels with one 16-byte pad per 32 labels on the way out
__device__ __forceinline__ uint32_t chunk_pos(uint32_t col, uint32_t rd) {
    const uint32_t cg = col >> 5, k = col & 31u;
    return cg * kCgStride + k * 16u + (((rd >> 2) ^ ((k >> 1) & 3u)) << 2) + (rd & 3u);
}

// What a lane (= one column key) keeps for the whole kernel: the counter blocks of a column are (0, 0, j_hi, j_lo), so
// after the first AddRoundKey the state columns 0 and 1 are the key words themselves — half of round 1's sixteen
// look-ups have constant inputs.  p[c] = rk_1[c] ^ (those two constant terms of column c); z, w = rk_0 words 2, 3.
struct CtrKey {
    uint32_t p[4], z, w;
    uint32_t q[4];  // p plus the terms of state column 2, valid while the counter's word 2 is zero (< 2^32 blocks)
};
__device__ __forceinline__ CtrKey ctr_key_setup(uint32_t keyaddr, uint32_t lo0) {
    const uint32_t lo2 = lo0 + 128u;
    const uint32_t sel0 = GC_PERM_SEL(0), sel1 = GC_PERM_SEL(1), sel2 = GC_PERM_SEL(2), sel3 = GC_PERM_SEL(3);
    const uint4 k0 = lds_ld4(keyaddr), k1 = lds_ld4(keyaddr + 16u);
    const uint32_t a0 = k0.x, a1 = k0.y;
    CtrKey c;
    // column c = Te0[b3(a_c)] ^ Te1[b2(a_c+1)] ^ Te2[b1(a_c+2)] ^ Te3[b0(a_c+3)], Te1 / Te3 = rotr8(Te0 / Te2)
    c.p[0] = k1.x ^ te_dual(a0, sel3, lo0) ^ rotr32(te_dual(a1, sel2, lo0), 8);
    c.p[1] = k1.y ^ te_dual(a1, sel3, lo0) ^ rotr32(te_dual(a0, sel0, lo2), 8);
    c.p[2] = k1.z ^ te_dual(a0, sel1, lo2) ^ rotr32(te_dual(a1, sel0, lo2), 8);
    c.p[3] = k1.w ^ te_dual(a1, sel1, lo2) ^ rotr32(te_dual(a0, sel2, lo0), 8);
    c.z = k0.z;
    c.w = k0.w;
    const uint32_t a2 = k0.z;  // j_hi = 0
    c.q[0] = c.p[0] ^ te_dual(a2, sel1, lo2);
    c.q[1] = c.p[1] ^ rotr32(te_dual(a2, sel2, lo0), 8);
    c.q[2] = c.p[2] ^ te_dual(a2, sel3, lo0);
    c.q[3] = c.p[3] ^ rotr32(te_dual(a2, sel0, lo2), 8);
    return c;
}

// N AES-128-CTR blocks in lock-step with a per-lane key schedule in LDS (keyaddr = byte address of round key 0):
// in: s[n][2], s[n][3] = the counter words j_hi, j_lo (words 0 and 1 of the block are zero); out: the four columns
// HI0: every counter of the launch is below 2^32 (launch-uniform): column 2 is constant as well, four look-ups
template <int N, bool HI0>
__device__ __forceinline__ void aes128_lanekey(uint32_t (&s)[N][4], uint32_t keyaddr, uint32_t lo0, const CtrKey &ck) {
    const uint32_t lo2 = lo0 + 128u;
    const uint32_t sel0 = GC_PERM_SEL(0), sel1 = GC_PERM_SEL(1), sel2 = GC_PERM_SEL(2), sel3 = GC_PERM_SEL(3);
    uint4 k;
    if constexpr (HI0) {  // round 1 from state column 3 alone
        uint32_t ad[N][4], t[N][4];
#pragma unroll
        for (int n = 0; n < N; n++) {
            const uint32_t a3 = s[n][3] ^ ck.w;
            ad[n][0] = __builtin_amdgcn_perm(a3, lo2, sel0);  // col 0: Te3[b0(a3)] (rotr8)
            ad[n][1] = __builtin_amdgcn_perm(a3, lo2, sel1);  // col 1: Te2[b1(a3)]
            ad[n][2] = __builtin_amdgcn_perm(a3, lo0, sel2);  // col 2: Te1[b2(a3)] (rotr8)
            ad[n][3] = __builtin_amdgcn_perm(a3, lo0, sel3);  // col 3: Te0[b3(a3)]
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < N; n++)
#pragma unroll
            for (int i = 0; i < 4; i++) t[n][i] = *(lds_u32 *)(uintptr_t)ad[n][i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < N; n++) {
            s[n][0] = ck.q[0] ^ rotr32(t[n][0], 8);
            s[n][1] = ck.q[1] ^ t[n][1];
            s[n][2] = ck.q[2] ^ rotr32(t[n][2], 8);
            s[n][3] = ck.q[3] ^ t[n][3];
        }
    } else {  // round 1: only the terms that depend on the counter (state columns 2 and 3): eight look-ups, not sixteen
        uint32_t ad[N][8], t[N][8];
#pragma unroll
        for (int n = 0; n < N; n++) {
            const uint32_t a2 = s[n][2] ^ ck.z, a3 = s[n][3] ^ ck.w;
            ad[n][0] = __builtin_amdgcn_perm(a2, lo2, sel1);  // col 0: Te2[b1(a2)]
            ad[n][1] = __builtin_amdgcn_perm(a3, lo2, sel0);  //        Te3[b0(a3)] (rotr8)
            ad[n][2] = __builtin_amdgcn_perm(a3, lo2, sel1);  // col 1: Te2[b1(a3)]
            ad[n][3] = __builtin_amdgcn_perm(a2, lo0, sel2);  //        Te1[b2(a2)] (rotr8)
            ad[n][4] = __builtin_amdgcn_perm(a2, lo0, sel3);  // col 2: Te0[b3(a2)]
            ad[n][5] = __builtin_amdgcn_perm(a3, lo0, sel2);  //        Te1[b2(a3)] (rotr8)
            ad[n][6] = __builtin_amdgcn_perm(a3, lo0, sel3);  // col 3: Te0[b3(a3)]
            ad[n][7] = __builtin_amdgcn_perm(a2, lo2, sel0);  //        Te3[b0(a2)] (rotr8)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < N; n++)
#pragma unroll
            for (int i = 0; i < 8; i++) t[n][i] = *(lds_u32 *)(uintptr_t)ad[n][i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < N; n++)
#pragma unroll
            for (int c = 0; c < 4; c++) s[n][c] = xor3(ck.p[c], t[n][2 * c], rotr32(t[n][2 * c + 1], 8));
    }
#pragma unroll
    for (int r = 2; r < 10; r++) {
        k = lds_ld4(keyaddr + 16u * r);
        const uint32_t kk[4] = {k.x, k.y, k.z, k.w};
        uint32_t ad[N][16], t[N][16];  // one batch of lookups per round (see aes_encrypt_dual)
#pragma unroll
        for (int n = 0; n < N; n++) te_round_addrs(s[n], lo0, lo2, ad[n]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < N; n++)
#pragma unroll
            for (int i = 0; i < 16; i++) t[n][i] = *(lds_u32 *)(uintptr_t)ad[n][i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < N; n++)
#pragma unroll
            for (int c = 0; c < 4; c++)
                s[n][c] = xor3(t[n][4 * c], t[n][4 * c + 1], kk[c]) ^ rotr32(t[n][4 * c + 2] ^ t[n][4 * c + 3], 8);
    }
    k = lds_ld4(keyaddr + 160u);
    {   // last round as in aes_encrypt_dual: one batch of 16 lookups per block, bytes merged with three bit-selects
        const uint32_t kk[4] = {k.x, k.y, k.z, k.w};
        uint32_t ad[N][16], t[N][16];
#pragma unroll
        for (int n = 0; n < N; n++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                ad[n][4 * c + 0] = __builtin_amdgcn_perm(s[n][c], lo2, sel3);
                ad[n][4 * c + 1] = __builtin_amdgcn_perm(s[n][(c + 1) & 3], lo0, sel2);
                ad[n][4 * c + 2] = __builtin_amdgcn_perm(s[n][(c + 2) & 3], lo0, sel1);
                ad[n][4 * c + 3] = __builtin_amdgcn_perm(s[n][(c + 3) & 3], lo2, sel0);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < N; n++)
#pragma unroll
            for (int i = 0; i < 16; i++) t[n][i] = *(lds_u32 *)(uintptr_t)ad[n][i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < N; n++)
#pragma unroll
            for (int c = 0; c < 4; c++) {  // byte 3 of t0, byte 2 of t1, byte 1 of t2, byte 0 of t3 (0xCA = m ? a : b)
                const uint32_t hi = __builtin_amdgcn_bitop3_b32(0xff000000u, t[n][4 * c], t[n][4 * c + 1], 0xCA);
                const uint32_t lo = __builtin_amdgcn_bitop3_b32(0x0000ff00u, t[n][4 * c + 2], t[n][4 * c + 3], 0xCA);
                s[n][c] = __builtin_amdgcn_bitop3_b32(0xffff0000u, hi, lo, 0xCA) ^ kk[c];
            }
    }
}

// 64 keystream bytes of one column starting at stream byte position p (p mod 16 == sh, launch-uniform):
// t[0..15] little-endian dwords.  prg() of iknp.go:632-637 restated for a lane.
template <bool MISALIGNED, bool HI0>
__device__ __forceinline__ void column_stream(uint64_t p, uint32_t sh, uint32_t keyaddr, uint32_t lo0, const CtrKey &ck,
                                              uint32_t (&t)[16]) {
    const uint64_t j0 = p >> 4;
    uint32_t w[20];
    // two blocks at a time: four in lock-step spill (1024-thread workgroups have 128 VGPRs per lane)
#pragma unroll
    for (int h = 0; h < 2; h++) {
        uint32_t s[2][4];
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const uint64_t j = j0 + 2 * h + n;
            s[n][0] = 0;
            s[n][1] = 0;
            s[n][2] = (uint32_t)(j >> 32);
            s[n][3] = (uint32_t)j;
        }
        aes128_lanekey<2, HI0>(s, keyaddr, lo0, ck);
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int c = 0; c < 4; c++) w[8 * h + 4 * n + c] = __builtin_bswap32(s[n][c]);
    }
    if (!MISALIGNED) {
#pragma unroll
        for (int i = 0; i < 16; i++) t[i] = w[i];
        return;
    }
    uint32_t e[1][4];
    const uint64_t j = j0 + 4;
    e[0][0] = 0;
    e[0][1] = 0;
    e[0][2] = (uint32_t)(j >> 32);
    e[0][3] = (uint32_t)j;
    aes128_lanekey<1, HI0>(e, keyaddr, lo0, ck);
#pragma unroll
    for (int c = 0; c < 4; c++) w[16 + c] = __builtin_bswap32(e[0][c]);
    // bytes [sh, sh + 64) of the 80 bytes: sh is wave-uniform, 1..15
    const uint32_t ws = sh >> 2, bs = (sh & 3u) * 8u;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {  // ws in 0..3
            a = ws == (uint32_t)d ? w[i + d] : a;
            b = ws == (uint32_t)d ? w[i + d + 1] : b;
        }
        t[i] = bs ? (a >> bs) | (b << (32u - bs)) : a;
    }
}

__device__ __forceinline__ uint32_t load_u8s(const uint8_t *src, uint32_t nbytes) {  // up to 4 bytes, little-endian
    uint32_t v = 0;
    for (uint32_t b = 0; b < nbytes; b++) v |= (uint32_t)src[b] << (8 * b);
    return v;
}

// bytes [16q, 16q+16) of a column of byte_rows (<= 64) bytes: full chunks are 16-byte aligned (col * 64), the last,
// shorter chunk of a call goes byte by byte
__device__ __forceinline__ uint4 load_quarter(const uint8_t *src, uint32_t byte_rows, uint32_t q) {
    if (byte_rows == 64) return ((const uint4 *)src)[q];
    uint32_t v[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        const uint32_t o = 16u * q + 4u * i;
        v[i] = o < byte_rows ? load_u8s(src + o, byte_rows - o < 4 ? byte_rows - o : 4) : 0;
    }
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store_quarter(uint8_t *dst, uint32_t byte_rows, uint32_t q, uint4 x) {
    if (byte_rows == 64) {
        ((uint4 *)dst)[q] = x;
        return;
    }
    const uint32_t v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
        for (uint32_t b = 0; b < 4; b++)
            if (16u * q + 4u * i + b < byte_rows) dst[16 * q + 4 * i + b] = (uint8_t)(v[i] >> (8 * b));
}

template <bool RECV, bool MISALIGNED, bool HI0>
__global__ __launch_bounds__(IKT) void k_iknp_fused(const uint32_t *__restrict__ rk0, const uint32_t *__restrict__ rk1,
                                                    uint64_t pos0, size_t n, const uint8_t *__restrict__ bbuf,
                                                    const uint8_t *__restrict__ u_in, uint4 delta,
                                                    uint8_t *__restrict__ u_out, uint4 *__restrict__ labels,
                                                    const uint32_t *__restrict__ g_te0) {
    extern __shared__ uint4 smem[];
    constexpr uint32_t NCH = RECV ? 4 : 8;                    // chunks per workgroup step
    constexpr uint32_t kKey0 = kTeDualBytes, kKey1 = kKey0 + kKeyBytes;
    constexpr uint32_t kBuf = RECV ? kKey1 + kKeyBytes : kKey1;  // byte address of chunk buffer 0
    load_te_dual((uint32_t *)smem, g_te0);
    for (uint32_t i = threadIdx.x; i < kKeyBytes / 4; i += IKT) {
        lds_st1(kKey0 + 4 * i, rk0[i]);
        if (RECV) lds_st1(kKey1 + 4 * i, rk1[i]);
    }
    __syncthreads();
    const uint32_t lo0 = te_lane_off();
    const uint32_t slice = threadIdx.x >> 7, col = threadIdx.x & 127u;
    const uint32_t cig = RECV ? (slice & 3u) : slice;       // chunk inside the group
    const uint32_t stream = RECV ? (slice >> 2) : 0u;       // receiver: 0 = g0 (t), 1 = g1
    const uint32_t keyaddr = (stream ? kKey1 : kKey0) + col * 176u;
    const uint32_t bufaddr = kBuf + cig * kChunkBuf;
    const CtrKey ck = ctr_key_setup(keyaddr, lo0);  // table and keys are in LDS (barrier above)
    const uint32_t sh = (uint32_t)(pos0 & 15u);
    const size_t chunks = (n + 511) / 512;
    const size_t groups = (chunks + NCH - 1) / NCH;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;

    for (size_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const size_t chunk = grp * NCH + cig;
        const bool valid = chunk < chunks;
        const size_t ofs = chunk * 512;
        const uint32_t rows = valid ? (uint32_t)((n - ofs) < 512 ? (n - ofs) : 512) : 0;
        const uint32_t byte_rows = (rows + 7) / 8;
        uint32_t t[16];
        if (valid) column_stream<MISALIGNED, HI0>(pos0 + 64 * (uint64_t)chunk, sh, keyaddr, lo0, ck, t);
        const size_t at = chunk * 8192 + (size_t)col * byte_rows;  // column-major message layout (iknp.go:490-499)
        if (RECV) {
            if (valid && stream == 1) {
#pragma unroll
                for (uint32_t q = 0; q < 4; q++)
                    lds_st4(bufaddr + 4 * chunk_pos(col, 4 * q), make_uint4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]));
            }
            __syncthreads();
            if (valid && stream == 0) {
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const uint4 b = load_quarter(bbuf + ofs / 8, byte_rows, q);  // choice bytes, same for every column
                    const uint4 t1 = lds_ld4(bufaddr + 4 * chunk_pos(col, 4 * q));
                    store_quarter(u_out + at, byte_rows, q,
                                  make_uint4(t[4 * q] ^ t1.x ^ b.x, t[4 * q + 1] ^ t1.y ^ b.y, t[4 * q + 2] ^ t1.z ^ b.z,
                                             t[4 * q + 3] ^ t1.w ^ b.w));
                }
            }
        } else if (valid) {
            // Delta.Bit(i): bit i of D0 for i < 64 (label.go:129-141) — D0 is the low limb
            const uint32_t word = col < 32 ? delta.x : col < 64 ? delta.y : col < 96 ? delta.z : delta.w;
            if ((word >> (col & 31u)) & 1u) {
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const uint4 u = load_quarter(u_in + at, byte_rows, q);
                    t[4 * q] ^= u.x, t[4 * q + 1] ^= u.y, t[4 * q + 2] ^= u.z, t[4 * q + 3] ^= u.w;
                }
            }
        }
        if (valid && stream == 0) {
#pragma unroll
            for (uint32_t q = 0; q < 4; q++)
                lds_st4(bufaddr + 4 * chunk_pos(col, 4 * q), make_uint4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]));
        }
        __syncthreads();
        // createLabels: wave w transposes chunk w of the group; lane = (rd, cg): dword rd (OTs 32rd .. 32rd+31) of the
        // 32 columns of group cg
        if (wave < NCH && grp * NCH + wave < chunks) {
            const size_t wofs = (grp * NCH + wave) * 512;
            const uint32_t wrows = (uint32_t)((n - wofs) < 512 ? (n - wofs) : 512);
            const uint32_t wbuf = kBuf + wave * kChunkBuf;
            const uint32_t rd = lane >> 2, cg = lane & 3u;
            uint32_t base[4];
#pragma unroll
            for (uint32_t h = 0; h < 4; h++)
                base[h] = wbuf + 4 * (cg * kCgStride + (((rd >> 2) ^ h) << 2) + (rd & 3u));  // chunk_pos without 16 k
            uint32_t x[32];
#pragma unroll
            for (uint32_t k = 0; k < 32; k++) x[k] = lds_ld1(base[(k >> 1) & 3u] + 64u * k);
            // 32 x 32 bit transpose (masked swaps): afterwards x[i] bit k = column (cg*32+k), OT 32rd+i
            uint32_t m = 0x0000ffffu;
#pragma unroll
            for (uint32_t j = 16; j != 0; j >>= 1, m ^= m << j) {
#pragma unroll
                for (uint32_t k = 0; k < 32; k = ((k | j) + 1) & ~j) {
                    const uint32_t tt = ((x[k] >> j) ^ x[k | j]) & m;
                    x[k] ^= tt << j;
                    x[k | j] ^= tt;
                }
            }
            // labels back through the same buffer (all reads of this wave are done: a wave's DS operations execute
            // in order and every write depends on the lane's 32 reads); label r sits at 16 * (r + r / 32): the pad
            // per 32 labels keeps the dword stores conflict-free and every address a base plus an immediate
            const uint32_t wbase = wbuf + 16 * (33 * rd) + 4 * cg;
#pragma unroll
            for (uint32_t i = 0; i < 32; i++) lds_st1(wbase + 16 * i, x[i]);
            const uint32_t rbase = wbuf + 16 * (lane + (lane >> 5));
#pragma unroll
            for (uint32_t mth = 0; mth < 8; mth++) {
                const uint32_t r = lane + 64 * mth;
                const uint4 v = lds_ld4(rbase + 16 * 66 * mth);
                if (r < wrows) labels[wofs + r] = v;  // uint4 = {D0 lo, D0 hi, D1 lo, D1 hi}: columns 0-31, 32-63, ...
            }
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t launch_iknp_fused(bool recv, const uint32_t *rk0, const uint32_t *rk1, uint64_t pos0, size_t n,
                             const uint8_t *bbuf, const uint8_t *u_in, uint4 delta, uint8_t *u_out, uint4 *labels,
                             const uint32_t *te0, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t chunks = (n + 511) / 512;
    const uint32_t nch = recv ? 4 : 8;
    const size_t groups = (chunks + nch - 1) / nch;
    const unsigned grid = (unsigned)(groups < 256 ? groups : 256);
    const size_t lds = kTeDualBytes + (recv ? 2 : 1) * kKeyBytes + (size_t)nch * kChunkBuf;
    const bool mis = (pos0 & 15u) != 0;
    // every counter this launch encrypts below 2^32 blocks (64 GiB of keystream per column): the cheaper first round
    // (GC_IKNP_GENERIC=1 forces the general first round: the only way to exercise it short of 64 GiB of stream)
    const char *gen = getenv("GC_IKNP_GENERIC");
    const bool hi0 = ((pos0 >> 4) + 4 * (uint64_t)chunks + 8) < (1ull << 32) && !(gen && gen[0] == '1');
    hipError_t e = hipSuccess;
#define GC_IK(R, M, H)                                                                                              \
    do {                                                                                                            \
        e = hipFuncSetAttribute((const void *)k_iknp_fused<R, M, H>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                (int)lds);                                                                          \
        if (e == hipSuccess)                                                                                        \
            hipLaunchKernelGGL((k_iknp_fused<R, M, H>), dim3(grid), dim3(IKT), lds, s, rk0, rk1, pos0, n, bbuf,     \
                               u_in, delta, u_out, labels, te0);                                                    \
    } while (0)
#define GC_IK2(R, M)            \
    do {                        \
        if (hi0) GC_IK(R, M, true); \
        else GC_IK(R, M, false);    \
    } while (0)
    if (recv) {
        if (mis) GC_IK2(true, true);
        else GC_IK2(true, false);
    } else {
        if (mis) GC_IK2(false, true);
        else GC_IK2(false, false);
    }
#undef GC_IK2
#undef GC_IK
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace gc
