// aes_narrow_ubench.hip — latency of ONE AES-256 hash when a phase has only a handful of blocks (developer tool, round 2):
// the column-sliced form of the kernels (4 lanes per block, hash_col_whitened) against a byte-sliced form with SIXTEEN
// lanes per block (lane = 4 * column + row inside a row of 16: one look-up per lane and round, the MixColumns sum as two
// row rotations by 5 and 10 lanes, a quad broadcast).  One workgroup per CU, W waves each running `iters` dependent
// hashes; reports ns per hash (= the latency a narrow phase pays) for W = 1, 4.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../mpc_amd/csrc/aes_device.h"
#include "../mpc_amd/csrc/aes_host.h"
using namespace gc;

constexpr uint32_t kKeyTab = kTeDualBytes;

template <int CTRL>
__device__ __forceinline__ uint32_t dppx(uint32_t v) {  // v ^ dpp(v)
    return v ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// byte-sliced: lane16 = 4c + r holds column c (all four r lanes the same value); rkc[round] = round key word of column c
template <int NR>
__device__ __forceinline__ uint32_t hash_byte_sliced(uint32_t s0, const uint32_t (&rkc)[NR + 1], uint32_t lane_off, uint32_t sel,
                                                     uint32_t rot, uint32_t bmask) {
    uint32_t s = s0;
#pragma unroll
    for (int r = 1; r < NR; r++) {
        const uint32_t ad = __builtin_amdgcn_perm(s, lane_off, sel);
        uint32_t v = *(lds_u32 *)(uintptr_t)ad;
        v = __builtin_amdgcn_alignbit(v, v, rot);                 // Te1 / Te3 = rotr8 of Te0 / Te2 (odd rows)
        uint32_t a = dppx<0x125>(v);                              // row_ror:5
        a = dppx<0x12A>(a);                                       // row_ror:10  -> lanes 4c' hold column c'
        a = (uint32_t)__builtin_amdgcn_mov_dpp((int)a, 0x00, 0xf, 0xf, true);  // quad_perm [0,0,0,0]
        s = a ^ rkc[r];
    }
    const uint32_t ad = __builtin_amdgcn_perm(s, lane_off ^ 128u, sel);  // last round: the other table half carries S there
    uint32_t v = *(lds_u32 *)(uintptr_t)ad & bmask;
    uint32_t a = dppx<0x125>(v);
    a = dppx<0x12A>(a);
    a = (uint32_t)__builtin_amdgcn_mov_dpp((int)a, 0x00, 0xf, 0xf, true);
    return xor3(a, rkc[NR], s0);
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint32_t *rk, const uint32_t *te0, uint32_t *out, int iters) {
    extern __shared__ uint4 smem[];
    load_te_dual((uint32_t *)smem, te0);
    if (threadIdx.x < 60) {
        uint32_t kv = rk[threadIdx.x];
        if (threadIdx.x >= 56) kv ^= rk[threadIdx.x - 56];
        ((uint32_t *)smem)[kKeyTab / 4 + threadIdx.x] = kv;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, l16 = lane & 15;
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    if (MODE == 0) {  // column-sliced, 4 lanes per block (production)
        const uint32_t c = lane & 3, keyaddr = kKeyTab + (c << 2), lo = te_lane_off();
        for (int it = 0; it < iters; it++) s = hash_col_whitened<14>(s, keyaddr, lo);
    } else {  // byte-sliced, 16 lanes per block
        const uint32_t c = l16 >> 2, r = l16 & 3;
        uint32_t rkc[15];
#pragma unroll
        for (int i = 0; i < 15; i++) rkc[i] = ((uint32_t *)smem)[kKeyTab / 4 + 4 * i + c];
        const uint32_t lane_off = te_lane_off() + (r >= 2 ? 128u : 0u);
        const uint32_t sel = GC_PERM_SEL(3 - r), rot = (r & 1) ? 8u : 0u, bmask = 0xffu << (8 * (3 - r));
        for (int it = 0; it < iters; it++) s = hash_byte_sliced<14>(s, rkc, lane_off, sel, rot, bmask);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    AesKey key;
    uint8_t kb[32];
    for (int i = 0; i < 32; i++) kb[i] = i;
    aes_expand_key(kb, 32, &key);
    uint32_t *d_rk, *d_te, *d_out;
    hipMalloc(&d_rk, 240);
    hipMalloc(&d_te, 1024);
    hipMalloc(&d_out, 256 * 1024 * 4);
    hipMemcpy(d_rk, key.w, 240, hipMemcpyHostToDevice);
    hipMemcpy(d_te, aes_tables().te0, 1024, hipMemcpyHostToDevice);
    const size_t lds = kTeDualBytes + 256;
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 2; mode++)
        for (int threads : {64, 256, 832, 1024}) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), lds, 0, d_rk, d_te, d_out, iters);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), lds, 0, d_rk, d_te, d_out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s  %4d threads/CU: %.1f ns per hash (%.1f ns per round), %d blocks per wave\n",
                   mode ? "byte-sliced (16 lanes/block)  " : "column-sliced (4 lanes/block)", threads, ms * 1e6 / iters,
                   ms * 1e6 / iters / 14, mode ? 4 : 16);
        }
    return 0;
}
