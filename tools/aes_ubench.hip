// aes_ubench.hip — micro-benchmark of the device AES cores in isolation (developer tool, not shipped).
// Every wave runs ITER dependent AES-256 encryptions (ILP chains per lane) out of the LDS tables; the
// result is cycles per block per CU for W waves per CU.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../mpc_amd/csrc/aes_device.h"
#include "../mpc_amd/csrc/aes_host.h"

using namespace gc;

template <int ILP, int MODE>
__global__ void k_bench(const uint32_t *rk, const uint32_t *te0, uint4 *out, int iters) {
    extern __shared__ uint4 smem[];
    uint32_t *te = (uint32_t *)smem;
    if (MODE == 0) load_te_dual(te, te0);
    else if (MODE == 1) load_te_replicated(te, te0);
    else load_te_tables(te, te0);
    uint32_t rkr[60];
    load_round_keys<14>(rkr, rk);
    __syncthreads();
    const uint32_t lo = te_lane_off();
    uint32_t s[ILP][4];
    for (int j = 0; j < ILP; j++) {
        s[j][0] = threadIdx.x * 7 + j;
        s[j][1] = blockIdx.x;
        s[j][2] = 0x1234567 * j;
        s[j][3] = threadIdx.x;
    }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) aes_encrypt_dual<14, ILP>(s, rkr, te, lo);
        else if (MODE == 1) aes_encrypt_repl<14, ILP>(s, rk, te, lo);
        else aes_encrypt_n<14, ILP>(s, rk, te);
    }
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int j = 0; j < ILP; j++) acc = make_uint4(acc.x ^ s[j][0], acc.y ^ s[j][1], acc.z ^ s[j][2], acc.w ^ s[j][3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int ILP, int MODE>
void run(const char *name, int threads, int blocks_per_cu, const uint32_t *d_rk, const uint32_t *d_te, uint4 *d_out) {
    const int cus = 256, iters = 200;
    size_t lds = MODE == 0 ? 65536 : MODE == 1 ? 32768 : 4096;
    hipFuncSetAttribute((const void *)k_bench<ILP, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dim3 grid(cus * blocks_per_cu), block(threads);
    hipLaunchKernelGGL((k_bench<ILP, MODE>), grid, block, lds, 0, d_rk, d_te, d_out, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_bench<ILP, MODE>), grid, block, lds, 0, d_rk, d_te, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double blocks = (double)grid.x * threads * ILP * iters;
    double per_cu_cycles = ms * 1e-3 * 2.4e9 / (blocks / cus);
    printf("%-10s threads=%4d x%d ILP=%d: %.3f ms, %.2f Gblocks/s, %.2f cycles/block/CU @2.4GHz\n", name, threads,
           blocks_per_cu, ILP, ms, blocks / ms / 1e6, per_cu_cycles);
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
    hipMalloc(&d_out, 256 * 2 * 1024 * 16);
    hipMemcpy(d_rk, k.w, 240, hipMemcpyHostToDevice);
    hipMemcpy(d_te, aes_tables().te0, 1024, hipMemcpyHostToDevice);
    run<1, 0>("dual", 256, 1, d_rk, d_te, d_out);
    run<1, 0>("dual", 512, 1, d_rk, d_te, d_out);
    run<1, 0>("dual", 1024, 1, d_rk, d_te, d_out);
    run<1, 0>("dual", 1024, 2, d_rk, d_te, d_out);
    run<2, 0>("dual", 512, 1, d_rk, d_te, d_out);
    run<2, 0>("dual", 1024, 1, d_rk, d_te, d_out);
    run<4, 0>("dual", 256, 1, d_rk, d_te, d_out);
    run<4, 0>("dual", 1024, 1, d_rk, d_te, d_out);
    run<1, 1>("repl32k", 1024, 1, d_rk, d_te, d_out);
    run<1, 1>("repl32k", 1024, 2, d_rk, d_te, d_out);
    run<2, 1>("repl32k", 1024, 2, d_rk, d_te, d_out);
    run<1, 2>("classic4k", 1024, 1, d_rk, d_te, d_out);
    run<1, 2>("classic4k", 1024, 2, d_rk, d_te, d_out);
    run<4, 2>("classic4k", 256, 8, d_rk, d_te, d_out);
    return 0;
}
