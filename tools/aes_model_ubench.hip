// aes_model_ubench.hip — how do VALU work and LDS lookups of the T-table AES round share a CU? (developer tool)
// One AES-256-shaped chain per lane (13 full rounds of 4 columns x 4 lookups); VARIANT changes only the number
// of VALU instructions per column (results are only meaningful for variants 0 and 1).
//   0: 4 perm + 5 (xor,xor,alignbit,xor,xor key)                 = 9 VALU / column
//   7: as 0 with a pointer-based table (one more v_add per lookup) = 13           (the core before this finding)
//   1: 4 perm + 4 (bitop3 t0^t2^key, xor, alignbit, xor)         = 8   (production core)
//   2: 4 perm + 3 (no rotate)                                    = 7
//   3: 4 perm + 2 (two bitop3)                                   = 6
//   4: 1 perm + 2, 4 reads at immediate offsets                  = 3   (LDS-dominated)
//   5: variant 0 without the LDS reads (value = address)         = 9 VALU, 0 LDS
//   6: variant 4 with 8 reads per column                          = LDS only, twice the reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../mpc_amd/csrc/aes_device.h"
#include "../mpc_amd/csrc/aes_host.h"
using namespace gc;

#define X3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)

template <int V>
__device__ __forceinline__ uint32_t col(const uint32_t *te, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                        uint32_t key, uint32_t lo0, uint32_t lo2) {
    const uint32_t sel0 = GC_PERM_SEL(0), sel1 = GC_PERM_SEL(1), sel2 = GC_PERM_SEL(2), sel3 = GC_PERM_SEL(3);
    if (V == 4 || V == 6) {
        const uint32_t addr = __builtin_amdgcn_perm(c0 ^ c1, lo0, sel3) & 0xff1fu;
        const uint32_t *p = (const uint32_t *)((const char *)te + addr);
        uint32_t r = X3(p[0], p[8], key) ^ X3(p[16], p[24], 0u);
        if (V == 6) r ^= X3(p[32], p[40], p[48]) ^ p[56];
        return r;
    }
    uint32_t t0, t1, t2, t3;
    if (V == 7) {  // pointer-based form: one v_add per lookup
        t0 = *(const uint32_t *)((const char *)te + __builtin_amdgcn_perm(c0, lo0, sel3));
        t2 = *(const uint32_t *)((const char *)te + __builtin_amdgcn_perm(c2, lo2, sel1));
        t1 = *(const uint32_t *)((const char *)te + __builtin_amdgcn_perm(c1, lo0, sel2));
        t3 = *(const uint32_t *)((const char *)te + __builtin_amdgcn_perm(c3, lo2, sel0));
    } else if (V == 5) {
        t0 = __builtin_amdgcn_perm(c0, lo0, sel3);
        t2 = __builtin_amdgcn_perm(c2, lo2, sel1);
        t1 = __builtin_amdgcn_perm(c1, lo0, sel2);
        t3 = __builtin_amdgcn_perm(c3, lo2, sel0);
    } else {
        t0 = te_dual(c0, sel3, lo0);
        t2 = te_dual(c2, sel1, lo2);
        t1 = te_dual(c1, sel2, lo0);
        t3 = te_dual(c3, sel0, lo2);
    }
    if (V == 0 || V == 5 || V == 7) return (t0 ^ t2) ^ rotr32(t1 ^ t3, 8) ^ key;
    if (V == 1) return X3(t0, t2, key) ^ rotr32(t1 ^ t3, 8);
    if (V == 2) return X3(t0, t2, key) ^ (t1 ^ t3);
    return X3(X3(t0, t2, key), t1, t3);
}

// variant 8: production column formula, but all 16 addresses first, then all 16 loads, then the combines
// (sched_barrier keeps the compiler from splitting the round into two 8-lookup halves with a wait in between)
__device__ __forceinline__ void round_batched(const uint32_t *te, uint32_t &a0, uint32_t &a1, uint32_t &a2, uint32_t &a3,
                                              const uint32_t *rk4, uint32_t lo0, uint32_t lo2) {
    const uint32_t sel0 = GC_PERM_SEL(0), sel1 = GC_PERM_SEL(1), sel2 = GC_PERM_SEL(2), sel3 = GC_PERM_SEL(3);
    const uint32_t s[4] = {a0, a1, a2, a3};
    uint32_t ad[16];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        ad[4 * c + 0] = __builtin_amdgcn_perm(s[c], lo0, sel3);
        ad[4 * c + 1] = __builtin_amdgcn_perm(s[(c + 2) & 3], lo2, sel1);
        ad[4 * c + 2] = __builtin_amdgcn_perm(s[(c + 1) & 3], lo0, sel2);
        ad[4 * c + 3] = __builtin_amdgcn_perm(s[(c + 3) & 3], lo2, sel0);
    }
    __builtin_amdgcn_sched_barrier(0);
    uint32_t t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = *(lds_u32 *)(uintptr_t)ad[i];
    __builtin_amdgcn_sched_barrier(0);
    uint32_t n[4];
#pragma unroll
    for (int c = 0; c < 4; c++) n[c] = X3(t[4 * c], t[4 * c + 1], rk4[c]) ^ rotr32(t[4 * c + 2] ^ t[4 * c + 3], 8);
    a0 = n[0], a1 = n[1], a2 = n[2], a3 = n[3];
}

template <int V>
__global__ __launch_bounds__(1024) void k_bench(const uint32_t *rk, const uint32_t *te0, uint4 *out, int iters) {
    extern __shared__ uint4 smem[];
    uint32_t *te = (uint32_t *)smem;
    load_te_dual(te, te0);
    uint32_t rkr[60];
    load_round_keys<14>(rkr, rk);
    __syncthreads();
    const uint32_t lo0 = te_lane_off(), lo2 = lo0 + 128;
    uint32_t a0 = threadIdx.x * 7, a1 = blockIdx.x, a2 = 0x1234567, a3 = threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 1; r < 14; r++) {
            if (V == 8) {
                round_batched(te, a0, a1, a2, a3, &rkr[4 * r], lo0, lo2);
                continue;
            }
            const uint32_t n0 = col<V>(te, a0, a1, a2, a3, rkr[4 * r + 0], lo0, lo2);
            const uint32_t n1 = col<V>(te, a1, a2, a3, a0, rkr[4 * r + 1], lo0, lo2);
            const uint32_t n2 = col<V>(te, a2, a3, a0, a1, rkr[4 * r + 2], lo0, lo2);
            const uint32_t n3 = col<V>(te, a3, a0, a1, a2, rkr[4 * r + 3], lo0, lo2);
            a0 = n0, a1 = n1, a2 = n2, a3 = n3;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = make_uint4(a0, a1, a2, a3);
}

template <int V>
void run(int threads, const uint32_t *d_rk, const uint32_t *d_te, uint4 *d_out) {
    const int cus = 256, iters = 200;
    hipFuncSetAttribute((const void *)k_bench<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_bench<V>), dim3(cus), dim3(threads), 65536, 0, d_rk, d_te, d_out, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_bench<V>), dim3(cus), dim3(threads), 65536, 0, d_rk, d_te, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per ROUND per CU for all waves of the CU, and per wave-round
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 13.0);
    printf("variant %d waves/CU=%2d: %.3f ms  %.0f cycles per round (all waves)  %.1f per wave-round  %.2f per lookup-instr\n",
           V, threads / 64, ms, cyc, cyc / (threads / 64), cyc / (threads / 64) / 16);
}

int main() {
    AesKey k;
    uint8_t key[32];
    for (int i = 0; i < 32; i++) key[i] = i;
    aes_expand_key(key, 32, &k);
    uint32_t *d_rk, *d_te;
    uint4 *d_out;
    hipMalloc(&d_rk, 60 * 4);
    hipMalloc(&d_te, 1024);
    hipMalloc(&d_out, 256 * 1024 * 16);
    hipMemcpy(d_rk, k.w, 240, hipMemcpyHostToDevice);
    hipMemcpy(d_te, aes_tables().te0, 1024, hipMemcpyHostToDevice);
    for (int threads : {64, 128, 256, 512, 576, 1024}) {
        run<0>(threads, d_rk, d_te, d_out);
        run<1>(threads, d_rk, d_te, d_out);
        run<2>(threads, d_rk, d_te, d_out);
        run<3>(threads, d_rk, d_te, d_out);
        run<4>(threads, d_rk, d_te, d_out);
        run<5>(threads, d_rk, d_te, d_out);
        run<6>(threads, d_rk, d_te, d_out);
        run<7>(threads, d_rk, d_te, d_out);
        run<8>(threads, d_rk, d_te, d_out);
    }
    return 0;
}
