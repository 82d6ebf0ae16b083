// ot_kernels.hip — CDNA4 kernels for the IKNP OT extension and MITCCRH.
//
//   k_iknp_prg        column AES-128-CTR PRG (+ u-matrix / delta fold)   ot/iknp.go:490-498, :214-219, :622-637
//   k_iknp_transpose  128 x w bit-matrix transpose -> labels (createLabels) ot/iknp.go:647-683
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

// keystream block j of a column: AES-128_rk(BE128(j)) as 16 stream bytes packed little-endian
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

// bytes [sh, sh+16) of the 32-byte concatenation lo || hi
__device__ __forceinline__ uint4 shift_bytes(uint4 lo, uint4 hi, uint32_t sh) {
    uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const uint32_t ws = sh >> 2, bs = (sh & 3) * 8;
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {  // register-only select of w[ws+k], w[ws+k+1]
            a = (uint32_t)t == ws + k ? w[t] : a;
            b = (uint32_t)t == ws + k + 1 ? w[t] : b;
        }
        o[k] = bs ? ((a >> bs) | (b << (32 - bs))) : a;
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ void store_bytes(uint8_t *dst, uint4 v, uint32_t nbytes) {
    if (nbytes == 16 && (((uintptr_t)dst) & 15) == 0) {
        *(uint4 *)dst = v;
        return;
    }
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    for (uint32_t t = 0; t < nbytes; t++) dst[t] = (uint8_t)(w[t >> 2] >> ((t & 3) * 8));
}

__device__ __forceinline__ uint4 load_bytes(const uint8_t *src, uint32_t nbytes) {
    if (nbytes == 16 && (((uintptr_t)src) & 15) == 0) return *(const uint4 *)src;
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t t = 0; t < nbytes; t++) w[t >> 2] |= (uint32_t)src[t] << ((t & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// One thread = one 16-byte segment of one column of one chunk.
//   RECV:  t = PRG(g0); u = PRG(g1) ^ t ^ choice-bytes   -> tbuf, u_out      (iknp.go:490-498)
//   SEND:  t = PRG(g0) ^ (delta bit ? u_in : 0)          -> tbuf             (iknp.go:214-219)
// grid.y = column (block-uniform round keys -> scalar loads)
template <bool RECV>
__global__ __launch_bounds__(256) void k_iknp_prg(const uint32_t *__restrict__ rk0, const uint32_t *__restrict__ rk1,
                                                  uint64_t pos0, size_t n, const uint8_t *__restrict__ bbuf,
                                                  const uint8_t *__restrict__ u_in, uint4 delta,
                                                  uint8_t *__restrict__ tbuf, uint8_t *__restrict__ u_out,
                                                  const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeWords];
    load_te_tables(te, g_te0);
    __syncthreads();
    const uint32_t col = blockIdx.y;
    const size_t seg = (size_t)blockIdx.x * 256 + threadIdx.x;  // global segment index: chunk*4 + q
    const size_t chunk = seg >> 2;
    const uint32_t q = (uint32_t)(seg & 3);
    const size_t ofs = chunk * 512;  // first OT of the chunk
    if (ofs >= n) return;
    const size_t rows = (n - ofs) < 512 ? (n - ofs) : 512;
    const uint32_t byte_rows = (uint32_t)((rows + 7) / 8);
    if (16 * q >= byte_rows) return;
    const uint32_t nbytes = byte_rows - 16 * q < 16 ? byte_rows - 16 * q : 16;
    // all earlier chunks of this call are full (64 bytes per column)
    const uint64_t p = pos0 + 64 * (uint64_t)chunk + 16 * q;  // stream byte position of this segment
    const uint32_t sh = (uint32_t)(p & 15);
    const uint64_t j0 = p >> 4;
    const uint32_t *k0 = rk0 + 44 * col;
    uint4 t;
    if (sh == 0) {  // launch-uniform: every segment of a call has the same misalignment
        uint64_t j[1] = {j0};
        uint4 o[1];
        ctr_blocks<1>(j, o, k0, te);
        t = o[0];
    } else {
        uint64_t j[2] = {j0, j0 + 1};
        uint4 o[2];
        ctr_blocks<2>(j, o, k0, te);
        t = shift_bytes(o[0], o[1], sh);
    }
    const size_t at = chunk * 8192 + (size_t)col * byte_rows + 16 * q;
    if (RECV) {
        const uint32_t *k1 = rk1 + 44 * col;
        uint4 t1;
        if (sh == 0) {
            uint64_t j[1] = {j0};
            uint4 o[1];
            ctr_blocks<1>(j, o, k1, te);
            t1 = o[0];
        } else {
            uint64_t j[2] = {j0, j0 + 1};
            uint4 o[2];
            ctr_blocks<2>(j, o, k1, te);
            t1 = shift_bytes(o[0], o[1], sh);
        }
        const uint4 b = load_bytes(bbuf + ofs / 8 + 16 * q, nbytes);
        store_bytes(tbuf + at, t, nbytes);
        store_bytes(u_out + at, lxor(lxor(t, t1), b), nbytes);
    } else {
        // Delta.Bit(i): bit i of D0 for i < 64 (label.go:129-141) — D0 is the LOW limb here
        const uint32_t word = col < 32 ? delta.x : col < 64 ? delta.y : col < 96 ? delta.z : delta.w;
        if ((word >> (col & 31)) & 1) t = lxor(t, load_bytes(u_in + at, nbytes));
        store_bytes(tbuf + at, t, nbytes);
    }
}

// createLabels (iknp.go:647-683): label 8*row+bit has bit j = bit `bit` of chunk[j*w + row];
// j < 64 -> D0 bit j, else D1 bit j-64.  One block per chunk: the chunk is staged in LDS with a
// padded column stride, every wave turns 8 byte-rows into 64 labels with wave-wide ballots
// (the 64-bit ballot of "bit b of column j's byte" IS limb D0 / D1 of label 8*row+b).
constexpr uint32_t kColStride = 68;  // bytes; 17 dwords -> conflict-free byte reads across columns

__global__ __launch_bounds__(256) void k_iknp_transpose(const uint8_t *__restrict__ tbuf, size_t n,
                                                        uint4 *__restrict__ labels) {
    __shared__ uint8_t lds[128 * kColStride];
    const size_t chunk = blockIdx.x;
    const size_t ofs = chunk * 512;
    if (ofs >= n) return;
    const size_t rows = (n - ofs) < 512 ? (n - ofs) : 512;
    const uint32_t w = (uint32_t)((rows + 7) / 8);
    const uint8_t *src = tbuf + chunk * 8192;
    for (uint32_t t = threadIdx.x; t < 128 * w; t += 256) {
        uint32_t col = t / w, r = t - col * w;
        lds[col * kColStride + r] = src[t];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t r0 = wave * 8; r0 < w; r0 += 32) {  // 8 byte-rows per wave pass
        uint64_t my0 = 0, my1 = 0;
#pragma unroll
        for (uint32_t rr = 0; rr < 8; rr++) {
            const uint32_t row = r0 + rr;
            uint32_t lo = 0, hi = 0;
            if (row < w) {
                lo = lds[lane * kColStride + row];
                hi = lds[(64 + lane) * kColStride + row];
            }
#pragma unroll
            for (uint32_t b = 0; b < 8; b++) {
                const uint64_t d0 = __ballot((lo >> b) & 1);
                const uint64_t d1 = __ballot((hi >> b) & 1);
                if (lane == rr * 8 + b) {
                    my0 = d0;
                    my1 = d1;
                }
            }
        }
        const size_t idx = (size_t)r0 * 8 + lane;  // label index inside the chunk
        if (idx < rows) labels[ofs + idx] = make_uint4((uint32_t)my0, (uint32_t)(my0 >> 32), (uint32_t)my1, (uint32_t)(my1 >> 32));
    }
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

void launch_iknp_prg(bool recv, const uint32_t *rk0, const uint32_t *rk1, uint64_t pos0, size_t n,
                     const uint8_t *bbuf, const uint8_t *u_in, uint4 delta, uint8_t *tbuf, uint8_t *u_out,
                     const uint32_t *te0, hipStream_t s) {
    if (n == 0) return;
    const size_t chunks = (n + 511) / 512;
    dim3 grid((unsigned)((chunks * 4 + 255) / 256), 128);
    if (recv)
        hipLaunchKernelGGL(k_iknp_prg<true>, grid, dim3(256), 0, s, rk0, rk1, pos0, n, bbuf, u_in, delta, tbuf, u_out,
                           te0);
    else
        hipLaunchKernelGGL(k_iknp_prg<false>, grid, dim3(256), 0, s, rk0, rk1, pos0, n, bbuf, u_in, delta, tbuf,
                           u_out, te0);
}

void launch_iknp_transpose(const uint8_t *tbuf, size_t n, uint4 *labels, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_iknp_transpose, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, s, tbuf, n, labels);
}

void launch_pack_bits(const uint8_t *b, size_t n, uint8_t *out, hipStream_t s) {
    if (n == 0) return;
    const size_t nb = (n + 7) / 8;
    hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, b, n, out);
}

}  // namespace gc
