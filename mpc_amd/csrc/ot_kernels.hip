// ot_kernels.hip — CDNA4 kernels for the IKNP OT extension and MITCCRH.
//
//   (the IKNP PRG + transpose live in iknp_fused_kernels.hip)
//   k_mitccrh / k_cot_send / k_cot_recv                                    ot/mitccrh.go:70-128, ot/cot.go:155-232
//
// A chunk is the reference's 8 KiB message: 128 columns x byteRows bytes, column-major.  Every
// column i owns an AES-128-CTR stream (key = its base-OT label, counter = 128-bit big-endian,
// zero IV) whose position persists across chunks and calls; all columns advance in lock-step, so
// the stream position is one scalar per call.
#include <algorithm>
#include <cstdlib>

#include "aes_device.h"
#include "kernels.h"

namespace gc {

__device__ __forceinline__ uint32_t bswap32d(uint32_t v) { return __builtin_bswap32(v); }

// keystream block j of a stream: AES-128_rk(BE128(j)) as 16 stream bytes packed little-endian
template <int N>
__device__ __forceinline__ void ctr_blocks(const uint64_t (&j)[N], uint4 (&out)[N], const uint32_t *__restrict__ rk,
                                           const uint32_t *te) {
    uint32_t s[N][4];
#pragma unroll
    for (int k = 0; k < N; k++) {
        s[k][0] = 0;
        s[k][1] = 0;
        s[k][2] = (uint32_t)(j[k] >> 32);
        s[k][3] = (uint32_t)j[k];
    }
    aes_encrypt_n<10, N>(s, rk, te);
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = make_uint4(bswap32d(s[k][0]), bswap32d(s[k][1]), bswap32d(s[k][2]), bswap32d(s[k][3]));
}

// choice bools -> bytes, LSB first (iknp.go:472-477)
__global__ void k_pack_bits(const uint8_t *__restrict__ b, size_t n, uint8_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 8 >= n) return;
    uint32_t v = 0;
    for (uint32_t k = 0; k < 8; k++)
        if (i * 8 + k < n && b[i * 8 + k]) v |= 1u << k;
    out[i] = (uint8_t)v;
}

// ---- MITCCRH ---------------------------------------------------------------------------------

// S-box lookup through the LDS T-tables: S[x] is byte 1 of Te0[x]
__device__ __forceinline__ uint32_t sbox(const uint32_t *te, uint32_t x) { return (te[x] >> 8) & 0xff; }

__device__ __forceinline__ uint32_t subword_rot(const uint32_t *te, uint32_t w) {
    // SubWord(RotWord(w)), big-endian word
    return (sbox(te, (w >> 16) & 0xff) << 24) | (sbox(te, (w >> 8) & 0xff) << 16) | (sbox(te, w & 0xff) << 8) |
           sbox(te, w >> 24);
}

// AES-128 with a per-thread key, key schedule computed on the fly (FIPS-197 §5.2); N blocks share the key.
template <int N>
__device__ __forceinline__ void aes128_otf(uint32_t (&s)[N][4], uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3,
                                           const uint32_t *te) {
    const uint32_t *te0 = te, *te1 = te + 256, *te2 = te + 512, *te3 = te + 768;
#pragma unroll
    for (int b = 0; b < N; b++) {
        s[b][0] ^= k0;
        s[b][1] ^= k1;
        s[b][2] ^= k2;
        s[b][3] ^= k3;
    }
    uint32_t rcon = 1;
#pragma unroll
    for (int r = 1; r <= 10; r++) {
        k0 ^= subword_rot(te, k3) ^ (rcon << 24);
        k1 ^= k0;
        k2 ^= k1;
        k3 ^= k2;
        rcon = (rcon << 1) ^ ((rcon & 0x80) ? 0x11b : 0);
#pragma unroll
        for (int b = 0; b < N; b++) {
            uint32_t a0 = s[b][0], a1 = s[b][1], a2 = s[b][2], a3 = s[b][3];
            if (r < 10) {
                s[b][0] = te0[a0 >> 24] ^ te1[(a1 >> 16) & 0xff] ^ te2[(a2 >> 8) & 0xff] ^ te3[a3 & 0xff] ^ k0;
                s[b][1] = te0[a1 >> 24] ^ te1[(a2 >> 16) & 0xff] ^ te2[(a3 >> 8) & 0xff] ^ te3[a0 & 0xff] ^ k1;
                s[b][2] = te0[a2 >> 24] ^ te1[(a3 >> 16) & 0xff] ^ te2[(a0 >> 8) & 0xff] ^ te3[a1 & 0xff] ^ k2;
                s[b][3] = te0[a3 >> 24] ^ te1[(a0 >> 16) & 0xff] ^ te2[(a1 >> 8) & 0xff] ^ te3[a2 & 0xff] ^ k3;
            } else {
                s[b][0] = (sbox(te, a0 >> 24) << 24) ^ (sbox(te, (a1 >> 16) & 0xff) << 16) ^ (sbox(te, (a2 >> 8) & 0xff) << 8) ^ sbox(te, a3 & 0xff) ^ k0;
                s[b][1] = (sbox(te, a1 >> 24) << 24) ^ (sbox(te, (a2 >> 16) & 0xff) << 16) ^ (sbox(te, (a3 >> 8) & 0xff) << 8) ^ sbox(te, a0 & 0xff) ^ k1;
                s[b][2] = (sbox(te, a2 >> 24) << 24) ^ (sbox(te, (a3 >> 16) & 0xff) << 16) ^ (sbox(te, (a0 >> 8) & 0xff) << 8) ^ sbox(te, a1 & 0xff) ^ k2;
                s[b][3] = (sbox(te, a3 >> 24) << 24) ^ (sbox(te, (a0 >> 16) & 0xff) << 16) ^ (sbox(te, (a1 >> 8) & 0xff) << 8) ^ sbox(te, a2 & 0xff) ^ k3;
            }
        }
    }
}

// MITCCRH key of OT index gid: BE(Label{D0:gid, D1:0} ^ seed)   (mitccrh.go:70-82)
__device__ __forceinline__ void mitccrh_key(uint4 seed, uint64_t gid, uint32_t (&k)[4]) {
    const uint64_t d0 = (((uint64_t)seed.y << 32) | seed.x) ^ gid;
    k[0] = (uint32_t)(d0 >> 32);
    k[1] = (uint32_t)d0;
    k[2] = seed.w;
    k[3] = seed.z;
}

// N labels hashed under one key: x ^= AES_key(x)  (mitccrh.go:107-127)
template <int N>
__device__ __forceinline__ void mitccrh_hash_n(uint4 (&x)[N], const uint32_t (&k)[4], const uint32_t *te) {
    uint32_t s[N][4];
#pragma unroll
    for (int b = 0; b < N; b++) {
        s[b][0] = x[b].y;
        s[b][1] = x[b].x;
        s[b][2] = x[b].w;
        s[b][3] = x[b].z;
    }
    aes128_otf<N>(s, k[0], k[1], k[2], k[3], te);
#pragma unroll
    for (int b = 0; b < N; b++) x[b] = lxor(x[b], cols_to_label(s[b]));
}

__global__ __launch_bounds__(256) void k_mitccrh(uint4 seed, uint64_t gid0, uint4 *__restrict__ blks, size_t n,
                                                 uint32_t h, const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeWords];
    load_te_tables(te, g_te0);
    __syncthreads();
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    uint32_t k[4];
    mitccrh_key(seed, gid0 + j, k);
    for (uint32_t t = 0; t < h; t++) {
        uint4 x[1] = {blks[j * h + t]};
        mitccrh_hash_n<1>(x, k, te);
        blks[j * h + t] = x[0];
    }
}

void launch_mitccrh_classic(uint4 seed, uint64_t gid0, uint4 *blks, size_t n, uint32_t h, const uint32_t *te0, hipStream_t s) {
    if (n == 0 || h == 0) return;
    hipLaunchKernelGGL(k_mitccrh, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, gid0, blks, n, h, te0);
}

// COT.Send pads (cot.go:160-181): out[2j] = H_j(x_j) ^ L0_j, out[2j+1] = H_j(x_j ^ delta) ^ L1_j
__global__ __launch_bounds__(256) void k_cot_send(uint4 seed, uint4 delta, const uint4 *__restrict__ data,
                                                  const uint4 *__restrict__ wires, size_t n, uint4 *__restrict__ out,
                                                  const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeWords];
    load_te_tables(te, g_te0);
    __syncthreads();
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    uint32_t k[4];
    mitccrh_key(seed, j, k);
    uint4 x[2] = {data[j], lxor(data[j], delta)};
    mitccrh_hash_n<2>(x, k, te);
    const uint4 z = make_uint4(0, 0, 0, 0);  // wires == nullptr: ROT (rot.go:168-171), the hashed pads themselves
    out[2 * j] = lxor(x[0], wires ? wires[2 * j] : z);
    out[2 * j + 1] = lxor(x[1], wires ? wires[2 * j + 1] : z);
}

void launch_cot_send_classic(uint4 seed, uint4 delta, const uint4 *data, const uint4 *wires, size_t n, uint4 *out,
                             const uint32_t *te0, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_cot_send, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, delta, data, wires, n,
                       out, te0);
}

// COT.Receive unpad (cot.go:203-232): result[j] = sent[2j + flag_j] ^ H_j(result[j])
__global__ __launch_bounds__(256) void k_cot_recv(uint4 seed, const uint8_t *__restrict__ flags,
                                                  const uint4 *__restrict__ sent, uint4 *__restrict__ result, size_t n,
                                                  const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeWords];
    load_te_tables(te, g_te0);
    __syncthreads();
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    uint32_t k[4];
    mitccrh_key(seed, j, k);
    uint4 x[1] = {result[j]};
    mitccrh_hash_n<1>(x, k, te);
    result[j] = lxor(x[0], sent[2 * j + (flags[j] ? 1 : 0)]);
}

void launch_cot_recv_classic(uint4 seed, const uint8_t *flags, const uint4 *sent, uint4 *result, size_t n,
                             const uint32_t *te0, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_cot_recv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, flags, sent, result, n,
                       te0);
}

// ---- MITCCRH / COT, tuned form -------------------------------------------------------------------------------
// Persistent 1024-thread workgroups around the perm-addressed dual table of the garbling kernels (conflict-free
// look-ups, one v_perm per address, 64 KiB at LDS offset 0; aes_device.h).  One lane = one OT: its AES-128 key is
// per-OT (mitccrh.go:70-89: key = BE(Label{gid, 0} ^ seed), renewed for every OT index), so the key schedule runs in the
// lane, one round ahead of the state and inside the SAME batch of LDS reads: round r's four SubWord look-ups depend
// only on round key r - 1 and are issued together with the 16 (or 32) state look-ups of round r.
// The first version (k_mitccrh / k_cot_send / k_cot_recv above, kept as GC_COT_CLASSIC=1) used a 4 KiB classic
// T-table per 256-thread block: ~3.5-way bank conflicts on every look-up and a table reload per 256 OTs.
constexpr int kCotThreads = 1024;

// SubWord(RotWord(w)) from four dual-table words: A = Te2[b2(w)], B = Te0[b1(w)], C = Te0[b0(w)], D = Te2[b3(w)]
// (S[x] is byte 3 and byte 0 of Te2[x], byte 2 and byte 1 of Te0[x]: the final-round selects of aes_encrypt_dual)
__device__ __forceinline__ uint32_t subrot_select(uint32_t A, uint32_t B, uint32_t C, uint32_t D) {
    const uint32_t hi = __builtin_amdgcn_bitop3_b32(0xff000000u, A, B, 0xCA);
    const uint32_t lo = __builtin_amdgcn_bitop3_b32(0x0000ff00u, C, D, 0xCA);
    return __builtin_amdgcn_bitop3_b32(0xffff0000u, hi, lo, 0xCA);
}

// N blocks under ONE per-lane AES-128 key k (big-endian words), key schedule on the fly; s in: plaintext columns,
// out: ciphertext columns
template <int N>
__device__ __forceinline__ void aes128_otf_dual(uint32_t (&s)[N][4], uint32_t (&k)[4], uint32_t lo0) {
    const uint32_t lo2 = lo0 + 128u;
    const uint32_t sel0 = GC_PERM_SEL(0), sel1 = GC_PERM_SEL(1), sel2 = GC_PERM_SEL(2), sel3 = GC_PERM_SEL(3);
#pragma unroll
    for (int b = 0; b < N; b++)
#pragma unroll
        for (int c = 0; c < 4; c++) s[b][c] ^= k[c];
    uint32_t rcon = 0x01000000u;
#pragma unroll
    for (int r = 1; r <= 10; r++) {
        uint32_t ad[N][16], t[N][16], ka[4], kt[4];
        // key schedule addresses: RotWord moves byte 2 to byte 3, 1 -> 2, 0 -> 1, 3 -> 0
        ka[0] = __builtin_amdgcn_perm(k[3], lo2, sel2);
        ka[1] = __builtin_amdgcn_perm(k[3], lo0, sel1);
        ka[2] = __builtin_amdgcn_perm(k[3], lo0, sel0);
        ka[3] = __builtin_amdgcn_perm(k[3], lo2, sel3);
        if (r < 10) {
#pragma unroll
            for (int b = 0; b < N; b++) te_round_addrs(s[b], lo0, lo2, ad[b]);
        } else {
#pragma unroll
            for (int b = 0; b < N; b++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    ad[b][4 * c + 0] = __builtin_amdgcn_perm(s[b][c], lo2, sel3);
                    ad[b][4 * c + 1] = __builtin_amdgcn_perm(s[b][(c + 1) & 3], lo0, sel2);
                    ad[b][4 * c + 2] = __builtin_amdgcn_perm(s[b][(c + 2) & 3], lo0, sel1);
                    ad[b][4 * c + 3] = __builtin_amdgcn_perm(s[b][(c + 3) & 3], lo2, sel0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; i++) kt[i] = *(lds_u32 *)(uintptr_t)ka[i];
#pragma unroll
        for (int b = 0; b < N; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) t[b][i] = *(lds_u32 *)(uintptr_t)ad[b][i];
        __builtin_amdgcn_sched_barrier(0);
        k[0] ^= subrot_select(kt[0], kt[1], kt[2], kt[3]) ^ rcon;
        k[1] ^= k[0];
        k[2] ^= k[1];
        k[3] ^= k[2];
        rcon = r == 8 ? 0x1b000000u : rcon << 1;
#pragma unroll
        for (int b = 0; b < N; b++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (r < 10) {
                    s[b][c] = xor3(t[b][4 * c], t[b][4 * c + 1], k[c]) ^ rotr32(t[b][4 * c + 2] ^ t[b][4 * c + 3], 8);
                } else {
                    s[b][c] = subrot_select(t[b][4 * c], t[b][4 * c + 1], t[b][4 * c + 2], t[b][4 * c + 3]) ^ k[c];
                }
            }
    }
}

// MODE 0: blks[j*h + t] ^= AES_key(j)(blks[j*h + t]), h = 1 or 2 (mitccrh.go:107-127)
// MODE 1: COT.Send pads   (cot.go:160-181): out[2j] = H_j(x_j) ^ L0_j, out[2j+1] = H_j(x_j ^ delta) ^ L1_j;
//         without wires: ROT.Send (rot.go:156-172): out[2j], out[2j+1] = the hashed pads = the sender's wire j
// MODE 2: COT.Receive     (cot.go:203-232): result[j] = sent[2j + flag_j] ^ H_j(result[j])
template <int MODE, int NB>
__global__ __launch_bounds__(kCotThreads) void k_cot_dual(uint4 seed, uint4 delta, uint64_t gid0, const uint4 *__restrict__ data,
                                                          const uint4 *__restrict__ wires, const uint8_t *__restrict__ flags,
                                                          uint4 *__restrict__ out, size_t n,
                                                          const uint32_t *__restrict__ g_te0) {
    extern __shared__ uint4 smem[];
    load_te_dual((uint32_t *)smem, g_te0);
    __syncthreads();
    const uint32_t lo0 = te_lane_off();
    for (size_t j = (size_t)blockIdx.x * kCotThreads + threadIdx.x; j < n; j += (size_t)gridDim.x * kCotThreads) {
        uint32_t k[4];
        mitccrh_key(seed, gid0 + j, k);
        uint4 x[NB];
        if (MODE == 0) {
#pragma unroll
            for (int b = 0; b < NB; b++) x[b] = out[j * NB + b];
        } else if (MODE == 1) {
            x[0] = data[j];
            if (NB > 1) x[NB - 1] = lxor(x[0], delta);
        } else {
            x[0] = out[j];
        }
        uint4 pad[NB];
        if (MODE == 1) {  // wires == nullptr: ROT.Send (rot.go:156-172) — the hashed pads ARE the sender's new wire labels
            const uint4 z = make_uint4(0, 0, 0, 0);
            pad[0] = wires ? wires[2 * j] : z;
            if (NB > 1) pad[NB - 1] = wires ? wires[2 * j + 1] : z;
        } else if (MODE == 2) {
            pad[0] = data[2 * j + (flags[j] ? 1 : 0)];  // data = the 2n labels received
        }
        uint32_t s[NB][4];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            s[b][0] = x[b].y;
            s[b][1] = x[b].x;
            s[b][2] = x[b].w;
            s[b][3] = x[b].z;
        }
        aes128_otf_dual<NB>(s, k, lo0);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            uint4 h = lxor(x[b], cols_to_label(s[b]));
            if (MODE != 0) h = lxor(h, pad[b]);
            if (MODE == 0) out[j * NB + b] = h;
            else if (MODE == 1) out[2 * j + b] = h;
            else out[j] = h;
        }
    }
}

static bool cot_classic() {
    static const bool v = std::getenv("GC_COT_CLASSIC") != nullptr;
    return v;
}

template <typename K>
static void launch_cot_dual(K kern, size_t n, hipStream_t s, uint4 seed, uint4 delta, uint64_t gid0, const uint4 *data,
                            const uint4 *wires, const uint8_t *flags, uint4 *out, const uint32_t *te0) {
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kTeDualBytes);
    const unsigned grid = (unsigned)std::min<size_t>(256, (n + kCotThreads - 1) / kCotThreads);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kCotThreads), kTeDualBytes, s, seed, delta, gid0, data, wires, flags, out, n, te0);
}

// ---- KOS consistency check (ot/iknp.go:138-194, 373-465; ot/gf128.go:14-27; ot/mul128_generic.go) --------
// acc[0..3] ^= XOR_i chi_i * v_i (256-bit carry-less product, no reduction), acc[4..5] ^= XOR_{bits_i} chi_i,
// chi_i = label (idx0 + i) of the AES-128-CTR stream keyed by seed2 (prgLabels, iknp.go:639-645).
// 128-bit values are little-endian word vectors here: for mul128 D0 is the LOW limb (mul128_generic.go:10-11),
// which is exactly the uint4 component order x,y,z,w.
// Persistent 1024-thread workgroups (dual AES table for the chi stream, round keys in SGPRs).  A lane multiplies its OTs
// one after the other and keeps XOR-ing the 256-bit products into ITS accumulator (the sums are linear: no reduction
// per OT); at the end the workgroup folds the lanes with wave shuffles + LDS and issues six 64-bit atomics — 256 x 6
// for a whole call.  (First version: one OT per thread of a 256-thread block, a wave reduction and six atomics per
// wave — 393 k same-address atomics for 4 Mi OTs — and a word loop with a dynamically indexed register array that
// the compiler spilled to scratch: 4.6 ms per 4 Mi OTs against 0.3 ms now.)
// The multiply is four 128 x 32-bit partial products (5 words each): per bit one mask, five (q ^= cur & m) as
// v_bitop3, five shifts — 352 ops per word, 1.4 k per OT instead of 2.3 k for the 8-word shift-and-add.
__global__ __launch_bounds__(kCotThreads) void k_kos_accumulate(const uint32_t *__restrict__ g_rk, uint64_t idx0,
                                                                const uint4 *__restrict__ v, const uint8_t *__restrict__ bits,
                                                                size_t n, unsigned long long *__restrict__ acc,
                                                                const uint32_t *__restrict__ g_te0) {
    extern __shared__ uint4 smem[];
    load_te_dual((uint32_t *)smem, g_te0);
    uint32_t rk[44];
    load_round_keys<10>(rk, g_rk);
    __syncthreads();
    const uint32_t lo0 = te_lane_off();
    uint32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xs[4] = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * kCotThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kCotThreads) {
        // chi_i = label (idx0 + i) of the AES-128-CTR stream: AES(BE128(idx0 + i)); the ciphertext columns ARE the
        // label's big-endian words: D0 = BE(bytes 0..7) -> (hi, lo) = (s0, s1), D1 -> (s2, s3)
        const uint64_t ctr = idx0 + i;
        uint32_t st[1][4] = {{0u, 0u, (uint32_t)(ctr >> 32), (uint32_t)ctr}};
        aes_encrypt_dual<10, 1, 0>(st, rk, (const uint32_t *)smem, lo0);
        const uint32_t chi[4] = {st[0][1], st[0][0], st[0][3], st[0][2]};  // little-endian word vector: D0 low, D0 high, D1 low, D1 high
        const uint4 b = v[i];
        const uint32_t bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int w = 0; w < 4; w++) {  // q = chi * bw[w] (128 x 32 -> 160 bits), added at word offset w
            uint32_t cur[5] = {chi[0], chi[1], chi[2], chi[3], 0u}, q[5] = {0u, 0u, 0u, 0u, 0u};
            uint32_t word = bw[w];
#pragma unroll 8
            for (int k = 0; k < 32; k++) {
                const uint32_t m = 0u - (word & 1u);
                word >>= 1;
#pragma unroll
                for (int t = 0; t < 5; t++) q[t] = __builtin_amdgcn_bitop3_b32(q[t], cur[t], m, 0x78);  // q ^ (cur & m)
#pragma unroll
                for (int t = 4; t > 0; t--) cur[t] = __builtin_amdgcn_alignbit(cur[t], cur[t - 1], 31);
                cur[0] <<= 1;
            }
#pragma unroll
            for (int t = 0; t < 5; t++) p[w + t] ^= q[t];
        }
        if (bits && bits[i]) {
#pragma unroll
            for (int t = 0; t < 4; t++) xs[t] ^= chi[t];
        }
    }
    // fold the lanes: wave shuffles, then the 16 waves through LDS (the table is no longer needed)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int q = 0; q < 8; q++) p[q] ^= __shfl_xor(p[q], off, 64);
#pragma unroll
        for (int q = 0; q < 4; q++) xs[q] ^= __shfl_xor(xs[q], off, 64);
    }
    __syncthreads();
    uint32_t *red = (uint32_t *)smem;
    if ((threadIdx.x & 63) == 0) {
        const uint32_t wv = threadIdx.x >> 6;
        for (int q = 0; q < 8; q++) red[wv * 12 + q] = p[q];
        for (int q = 0; q < 4; q++) red[wv * 12 + 8 + q] = xs[q];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        uint32_t r = 0;
        for (uint32_t wv = 0; wv < kCotThreads / 64; wv++) r ^= red[wv * 12 + threadIdx.x];
        red[256 + threadIdx.x] = r;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const unsigned long long val = ((unsigned long long)red[256 + 2 * threadIdx.x + 1] << 32) | red[256 + 2 * threadIdx.x];
        if (val) atomicXor(&acc[threadIdx.x], val);
    }
}

void launch_kos_accumulate(const uint32_t *rk, uint64_t idx0, const uint4 *v, const uint8_t *bits, size_t n,
                           unsigned long long *acc, const uint32_t *te0, hipStream_t s) {
    if (n == 0) return;
    (void)hipFuncSetAttribute((const void *)k_kos_accumulate, hipFuncAttributeMaxDynamicSharedMemorySize, kTeDualBytes);
    const unsigned grid = (unsigned)std::min<size_t>(256, (n + kCotThreads - 1) / kCotThreads);
    hipLaunchKernelGGL(k_kos_accumulate, dim3(grid), dim3(kCotThreads), kTeDualBytes, s, rk, idx0, v, bits, n, acc, te0);
}

void launch_mitccrh(uint4 seed, uint64_t gid0, uint4 *blks, size_t n, uint32_t h, const uint32_t *te0, hipStream_t s) {
    if (n == 0 || h == 0) return;
    const uint4 z = make_uint4(0, 0, 0, 0);
    if (cot_classic() || h > 2) return launch_mitccrh_classic(seed, gid0, blks, n, h, te0, s);
    if (h == 1) launch_cot_dual(k_cot_dual<0, 1>, n, s, seed, z, gid0, nullptr, nullptr, nullptr, blks, te0);
    else launch_cot_dual(k_cot_dual<0, 2>, n, s, seed, z, gid0, nullptr, nullptr, nullptr, blks, te0);
}

void launch_cot_send(uint4 seed, uint4 delta, const uint4 *data, const uint4 *wires, size_t n, uint4 *out,
                     const uint32_t *te0, hipStream_t s) {
    if (n == 0) return;
    if (cot_classic()) return launch_cot_send_classic(seed, delta, data, wires, n, out, te0, s);
    launch_cot_dual(k_cot_dual<1, 2>, n, s, seed, delta, 0, data, wires, nullptr, out, te0);
}

void launch_cot_recv(uint4 seed, const uint8_t *flags, const uint4 *sent, uint4 *result, size_t n,
                     const uint32_t *te0, hipStream_t s) {
    if (n == 0) return;
    if (cot_classic()) return launch_cot_recv_classic(seed, flags, sent, result, n, te0, s);
    launch_cot_dual(k_cot_dual<2, 1>, n, s, seed, make_uint4(0, 0, 0, 0), 0, sent, nullptr, flags, result, te0);
}

// ---- bit-COT helpers (gc_iknp_*_bits_dev) ------------------------------------------------------------------------
// choice words (packed little-endian u64, one bit per OT) -> the 64-byte-per-chunk choice buffer of the fused kernel,
// with the reference's fold of WHOLE 64-bit words only (iknp.go:583-597): a trailing partial word of the last chunk
// does not enter u
__global__ void k_fold_choice_words(const uint64_t *__restrict__ choices, size_t n, uint64_t *__restrict__ bbuf) {
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // word index, 8 per chunk
    const size_t chunks = (n + 511) / 512;
    if (w >= chunks * 8) return;
    const size_t c = w / 8, ofs = c * 512, rows = n - ofs < 512 ? n - ofs : 512, words = ((rows + 7) / 8) / 8;
    bbuf[w] = (w % 8) < words ? choices[w] : 0;
}
// result bit i = labels[i].Bit(0) (D0 bit 0), packed little-endian u64 (iknp.go:609-614, :299-306)
__global__ void k_pack_label_bit0(const uint4 *__restrict__ labels, size_t n, uint64_t *__restrict__ result) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bit = i < n && (labels[i].x & 1u);
    const uint64_t m = __ballot(bit);
    if ((threadIdx.x & 63) == 0 && (i & ~(size_t)63) < n) result[i / 64] = m;
}
void launch_fold_choice_words(const uint64_t *choices, size_t n, uint64_t *bbuf, hipStream_t s) {
    const size_t words = ((n + 511) / 512) * 8;
    if (words == 0) return;
    hipLaunchKernelGGL(k_fold_choice_words, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, choices, n, bbuf);
}
void launch_pack_label_bit0(const uint4 *labels, size_t n, uint64_t *result, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pack_label_bit0, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, labels, n, result);
}

// ---- IKNP launchers ----------------------------------------------------------------------------

void launch_pack_bits(const uint8_t *b, size_t n, uint8_t *out, hipStream_t s) {
    if (n == 0) return;
    const size_t nb = (n + 7) / 8;
    hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, b, n, out);
}

}  // namespace gc
