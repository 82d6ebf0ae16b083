// ot_kernels.hip — CDNA4 kernels for the IKNP OT extension and MITCCRH.
//
//   (the IKNP PRG + transpose live in iknp_fused_kernels.hip)
//   k_mitccrh / k_cot_send / k_cot_recv                                    ot/mitccrh.go:70-128, ot/cot.go:155-232
//
// A chunk is the reference's 8 KiB message: 128 columns x byteRows bytes, column-major.  Every
// column i owns an AES-128-CTR stream (key = its base-OT label, counter = 128-bit big-endian,
// zero IV) whose position persists across chunks and calls; all columns advance in lock-step, so
// the stream position is one scalar per call.
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

void launch_mitccrh(uint4 seed, uint64_t gid0, uint4 *blks, size_t n, uint32_t h, const uint32_t *te0, hipStream_t s) {
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
    out[2 * j] = lxor(x[0], wires[2 * j]);
    out[2 * j + 1] = lxor(x[1], wires[2 * j + 1]);
}

void launch_cot_send(uint4 seed, uint4 delta, const uint4 *data, const uint4 *wires, size_t n, uint4 *out,
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

void launch_cot_recv(uint4 seed, const uint8_t *flags, const uint4 *sent, uint4 *result, size_t n,
                     const uint32_t *te0, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_cot_recv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, flags, sent, result, n,
                       te0);
}

// ---- KOS consistency check (ot/iknp.go:138-194, 373-465; ot/gf128.go:14-27; ot/mul128_generic.go) --------
// acc[0..3] ^= XOR_i chi_i * v_i (256-bit carry-less product, no reduction), acc[4..5] ^= XOR_{bits_i} chi_i,
// chi_i = label (idx0 + i) of the AES-128-CTR stream keyed by seed2 (prgLabels, iknp.go:639-645).
// 128-bit values are little-endian word vectors here: for mul128 D0 is the LOW limb (mul128_generic.go:10-11),
// which is exactly the uint4 component order x,y,z,w.
__global__ __launch_bounds__(256) void k_kos_accumulate(const uint32_t *__restrict__ rk, uint64_t idx0,
                                                        const uint4 *__restrict__ v, const uint8_t *__restrict__ bits,
                                                        size_t n, unsigned long long *__restrict__ acc,
                                                        const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeWords];
    load_te_tables(te, g_te0);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xs[4] = {0, 0, 0, 0};
    if (i < n) {
        uint64_t j[1] = {idx0 + i};
        uint4 o[1];
        ctr_blocks<1>(j, o, rk, te);  // stream bytes, little-endian packed
        // Label.SetBytes: D0 = BE(bytes 0..7), D1 = BE(bytes 8..15)
        const uint4 chi = make_uint4(bswap32d(o[0].y), bswap32d(o[0].x), bswap32d(o[0].w), bswap32d(o[0].z));
        const uint4 b = v[i];
        // carry-less 128 x 128 -> 256: shift-and-add over the bits of b (clmul64 loop of mul128_generic.go:30-46)
        uint32_t cur[8] = {chi.x, chi.y, chi.z, chi.w, 0, 0, 0, 0};
        const uint32_t bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll 1
        for (int w = 0; w < 4; w++) {
            uint32_t word = bw[w];
#pragma unroll 4
            for (int k = 0; k < 32; k++) {
                const uint32_t m = 0u - (word & 1u);
                word >>= 1;
#pragma unroll
                for (int q = 0; q < 8; q++) p[q] ^= cur[q] & m;
#pragma unroll
                for (int q = 7; q > 0; q--) cur[q] = __builtin_amdgcn_alignbit(cur[q], cur[q - 1], 31);
                cur[0] <<= 1;
            }
        }
        if (bits && bits[i]) {
            xs[0] = chi.x;
            xs[1] = chi.y;
            xs[2] = chi.z;
            xs[3] = chi.w;
        }
    }
    // XOR-reduce over the wave, one 64-bit atomic per accumulator limb per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int q = 0; q < 8; q++) p[q] ^= __shfl_xor(p[q], off, 64);
#pragma unroll
        for (int q = 0; q < 4; q++) xs[q] ^= __shfl_xor(xs[q], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        for (int q = 0; q < 4; q++) atomicXor(&acc[q], ((unsigned long long)p[2 * q + 1] << 32) | p[2 * q]);
        for (int q = 0; q < 2; q++) atomicXor(&acc[4 + q], ((unsigned long long)xs[2 * q + 1] << 32) | xs[2 * q]);
    }
}

void launch_kos_accumulate(const uint32_t *rk, uint64_t idx0, const uint4 *v, const uint8_t *bits, size_t n,
                           unsigned long long *acc, const uint32_t *te0, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_kos_accumulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rk, idx0, v, bits, n, acc,
                       te0);
}

// ---- IKNP launchers ----------------------------------------------------------------------------

void launch_pack_bits(const uint8_t *b, size_t n, uint8_t *out, hipStream_t s) {
    if (n == 0) return;
    const size_t nb = (n + 7) / 8;
    hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, b, n, out);
}

}  // namespace gc
