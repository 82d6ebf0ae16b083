// issue_rate_ubench.hip — what does a gfx950 SIMD / CU really issue per cycle? (developer tool, round 2)
//
// The T-table AES core is 32 VALU + 16 ds_read_b32 per round and block.  To decide whether its measured rate is a
// hardware floor we need the machine's own numbers, not the model's:
//   (1) cycles per wave64 instruction of every VALU opcode the core uses (and the candidates that could replace
//       them), at 1 / 2 / 4 waves per SIMD;
//   (2) cycles per wave64 LDS read for the access shapes of interest (conflict-free dword, the production
//       perm-addressed replicated row, 8-byte stride, unaligned dword, byte / short reads, b64);
//   (3) how VALU and LDS issue overlap on one SIMD: 16 look-ups + N VALU per iteration for N = 0 ... 48, the VALU
//       independent of the loads (pure issue test), at 4 waves per SIMD, with and without s_setprio role separation;
//   (4) the shader clock during the run (s_memtime ticks per wall-clock second).
// Output: one line per experiment, cycles per wave-instruction per SIMD (VALU) or per CU (LDS).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define HIPCHECK(x)                                                                     \
    do {                                                                                \
        hipError_t e__ = (x);                                                           \
        if (e__ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__));                    \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

// 8 independent chains x 8 = 64 instructions per asm block
#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP8(X) X X X X X X X X

#define VALU_KERNEL(NAME, INS)                                                                                 \
    __global__ __launch_bounds__(1024) void NAME(uint32_t *out, int iters, uint64_t *ticks) {                 \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, \
                 a7 = a0 * 19;                                                                                 \
        uint32_t b = blockIdx.x * 77 + threadIdx.x, c = b * 31 + 5;                                            \
        uint32_t s = __builtin_amdgcn_readfirstlane(blockIdx.x * 13 + 1);                                      \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                      \
        for (int it = 0; it < iters; it++) {                                                                   \
            asm volatile(REP8(INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")) \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)       \
                         : "v"(b), "v"(c), "s"(s));                                                            \
        }                                                                                                      \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                    \
        if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;                                                     \
    }

#define I_XOR(A) "v_xor_b32 " A ", " A ", %8\n"
#define I_XOR_S(A) "v_xor_b32 " A ", %10, " A "\n"
#define I_ADD(A) "v_add_u32 " A ", " A ", %8\n"
#define I_PERM(A) "v_perm_b32 " A ", " A ", %8, %9\n"
#define I_ALIGN(A) "v_alignbit_b32 " A ", " A ", " A ", 8\n"
#define I_BITOP3(A) "v_bitop3_b32 " A ", " A ", %8, %9 bitop3:0x96\n"
#define I_BITOP3_S(A) "v_bitop3_b32 " A ", " A ", %8, %10 bitop3:0x96\n"
#define I_LSHLADD(A) "v_lshl_add_u32 " A ", " A ", 4, %8\n"
#define I_ANDOR(A) "v_and_or_b32 " A ", " A ", %8, %9\n"
#define I_BFE(A) "v_bfe_u32 " A ", " A ", 8, 8\n"
#define I_LSHL(A) "v_lshlrev_b32 " A ", 1, " A "\n"
#define I_LSHL_V(A) "v_lshlrev_b32 " A ", %9, " A "\n"
#define I_AND(A) "v_and_b32 " A ", " A ", %8\n"
#define I_OR(A) "v_or_b32 " A ", " A ", %8\n"
#define I_XOR_C(A) "v_xor_b32 " A ", 0x55, " A "\n"
#define I_MOVDPP(A) "v_mov_b32_dpp " A ", " A " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_XORDPP(A) "v_xor_b32_dpp " A ", " A ", " A " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_FMA(A) "v_fma_f32 " A ", " A ", %8, %9\n"
#define I_ADDF(A) "v_add_f32 " A ", " A ", %8\n"
#define I_PKADD16(A) "v_pk_add_u16 " A ", " A ", %8\n"
#define I_MOV(A) "v_mov_b32 " A ", %8\n"
#define I_SDWA(A) "v_lshlrev_b32_sdwa " A ", %9, " A " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"

VALU_KERNEL(k_xor, I_XOR)
VALU_KERNEL(k_xor_s, I_XOR_S)
VALU_KERNEL(k_add, I_ADD)
VALU_KERNEL(k_perm, I_PERM)
VALU_KERNEL(k_align, I_ALIGN)
VALU_KERNEL(k_bitop3, I_BITOP3)
VALU_KERNEL(k_bitop3_s, I_BITOP3_S)
VALU_KERNEL(k_lshladd, I_LSHLADD)
VALU_KERNEL(k_andor, I_ANDOR)
VALU_KERNEL(k_bfe, I_BFE)
VALU_KERNEL(k_lshl, I_LSHL)
VALU_KERNEL(k_lshl_v, I_LSHL_V)
VALU_KERNEL(k_and, I_AND)
VALU_KERNEL(k_or, I_OR)
VALU_KERNEL(k_xor_c, I_XOR_C)
VALU_KERNEL(k_movdpp, I_MOVDPP)
VALU_KERNEL(k_xordpp, I_XORDPP)
VALU_KERNEL(k_fma, I_FMA)
VALU_KERNEL(k_addf, I_ADDF)
VALU_KERNEL(k_pkadd16, I_PKADD16)
VALU_KERNEL(k_mov, I_MOV)
VALU_KERNEL(k_sdwa, I_SDWA)

// ---- LDS reads: 16 loads per batch, one s_waitcnt per batch -------------------------------------------------------
// MODE: address shape (see main); the 16 addresses of a batch differ by an XOR with the batch's previous data so that
// the loop cannot be hoisted, like the AES rounds' data-dependent indices.
#define LDS16(OP, OFFS)                                                                                       \
    asm volatile(OP " %0, %16" OFFS "\n" OP " %1, %17" OFFS "\n" OP " %2, %18" OFFS "\n" OP " %3, %19" OFFS "\n"  \
                 OP " %4, %20" OFFS "\n" OP " %5, %21" OFFS "\n" OP " %6, %22" OFFS "\n" OP " %7, %23" OFFS "\n"  \
                 OP " %8, %24" OFFS "\n" OP " %9, %25" OFFS "\n" OP " %10, %26" OFFS "\n" OP " %11, %27" OFFS "\n" \
                 OP " %12, %28" OFFS "\n" OP " %13, %29" OFFS "\n" OP " %14, %30" OFFS "\n" OP " %15, %31" OFFS "\n" \
                 "s_waitcnt lgkmcnt(0)\n"                                                                     \
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]),   \
                   "=&v"(t[7]), "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), \
                   "=&v"(t[14]), "=&v"(t[15])                                                                 \
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]), \
                   "v"(ad[8]), "v"(ad[9]), "v"(ad[10]), "v"(ad[11]), "v"(ad[12]), "v"(ad[13]), "v"(ad[14]),     \
                   "v"(ad[15])                                                                                \
                 : "memory")

template <int MODE, int NVALU, int PRIO>
__global__ __launch_bounds__(1024) void k_lds(uint32_t *out, int iters, uint64_t *ticks) {
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < 65536 / 4 + 64; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t ad[16], t[16];
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t v0 = x, v1 = x * 3, v2 = x * 5, v3 = x * 7, b = x ^ 0x55, c = x + 9;
    if (PRIO == 1 && ((threadIdx.x >> 6) & 4)) __builtin_amdgcn_s_setprio(1);
    // the 16 addresses of a lane are fixed (random per lane and slot): no address arithmetic inside the timed loop
#pragma unroll
    for (int i = 0; i < 16; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t idx = (x >> 24) & 0xffu;
        if (MODE == 0) ad[i] = lane * 4 + ((x >> 8) & 0xff00u);                // dword, bank = lane: conflict-free
        if (MODE == 1) ad[i] = (idx << 8) | ((lane & 31) * 4);                 // production: replicated row
        if (MODE == 2) ad[i] = (idx << 8) | ((lane & 31) * 4 + 128);           // production, second half
        if (MODE == 3) ad[i] = (idx << 8) | ((lane & 31) * 8);                 // 8-byte stride (2-way under mod 32)
        if (MODE == 4) ad[i] = ((idx << 8) | ((lane & 31) * 4)) + 1;           // unaligned dword
        if (MODE == 5) ad[i] = (idx << 8) | ((lane & 31) * 4);                 // u8 read
        if (MODE == 6) ad[i] = (idx << 8) | ((lane & 31) * 4);                 // u16 read
        if (MODE == 7) ad[i] = lane * 8 + ((x >> 8) & 0xfe00u);                // b64, linear
        if (MODE == 8) ad[i] = (idx << 8) | ((lane & 31) * 8);                 // b64 on the 8-byte-stride row
        if (MODE == 9) ad[i] = (idx << 8) | (lane * 4);                        // 64 copies: lane-private bank
    }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 5) LDS16("ds_read_u8", "");
        else if (MODE == 6) LDS16("ds_read_u16", "");
        else if (MODE == 7 || MODE == 8) {
            // b64: 16 loads into 16 register pairs
            uint64_t q[16];
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("ds_read_b64 %0, %1\n" : "=v"(q[i]) : "v"(ad[i]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; i++) t[i] = (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
        } else LDS16("ds_read_b32", "");
        // N VALU instructions that do not depend on the loads (issue test); they follow the loads in program order,
        // so inside one wave they can only overlap the loads' latency, across waves anything
        if (NVALU > 0) {
#pragma unroll
            for (int k = 0; k < NVALU / 4; k++)
                asm volatile("v_perm_b32 %0, %0, %4, %5\nv_bitop3_b32 %1, %1, %4, %5 bitop3:0x96\n"
                             "v_alignbit_b32 %2, %2, %2, 8\nv_xor_b32 %3, %3, %4\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)
                             : "v"(b), "v"(c));
        }
        x ^= t[0];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x ^ v0 ^ v1 ^ v2 ^ v3;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// ---- vector-memory gathers from a small table that stays in the CU's L1: is the texture path a second look-up engine?
// TABLE_BYTES = 1024 (compact Te0) or 65536 (the LDS layout); 16 loads per batch
template <int SHIFT>
__global__ __launch_bounds__(1024) void k_vmem(const uint32_t *tab, uint32_t *out, int iters, uint64_t *ticks) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t off[16], t[16];
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t idx = (x >> 24) & 0xffu;
        off[i] = SHIFT == 2 ? idx * 4 : ((idx << 8) | ((lane & 31) * 4));
    }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            asm volatile("global_load_dword %0, %1, %2" : "=v"(t[i]) : "v"(off[i]), "s"(tab) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        x ^= t[0];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static uint32_t *d_out;
static uint64_t *d_ticks;
static double g_clock_hz = 2.4e9;

static int g_grid = 256;
template <typename K>
static double run(K kern, int threads, size_t lds, int iters, double *tick_cycles) {
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    if (lds) HIPCHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(g_grid), dim3(threads), lds, 0, d_out, 4, d_ticks);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(g_grid), dim3(threads), lds, 0, d_out, iters, d_ticks);
    HIPCHECK(hipEventRecord(e1));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(g_grid);
    HIPCHECK(hipMemcpy(h.data(), d_ticks, g_grid * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += (double)v;
    *tick_cycles = avg / g_grid;
    return ms * 1e-3;
}

int main() {
    HIPCHECK(hipMalloc(&d_out, 512 * 1024 * 4));
    HIPCHECK(hipMalloc(&d_ticks, 512 * 8));
    {  // warm the clocks up before anything is measured
        double ticks;
        for (int i = 0; i < 5; i++) run(k_xor, 1024, 0, 20000, &ticks);
    }
    // (4) clock: ticks of s_memtime per wall second on a long VALU kernel (after the warm-up above)
    {
        double ticks;
        const double s = run(k_xor, 1024, 0, 20000, &ticks);
        printf("clock: s_memtime %.0f ticks in %.3f ms wall -> %.1f MHz (kernel incl. launch)\n", ticks, s * 1e3, ticks / s / 1e6);
        g_clock_hz = ticks / s;
    }
    // (1) VALU opcodes: 64 instructions per iteration per wave; waves per SIMD = threads / 256
    struct V {
        const char *name;
        void (*k)(uint32_t *, int, uint64_t *);
    } vs[] = {{"v_xor_b32 (vgpr,vgpr)", k_xor},   {"v_xor_b32 (sgpr,vgpr)", k_xor_s}, {"v_add_u32", k_add},
              {"v_perm_b32", k_perm},             {"v_alignbit_b32", k_align},        {"v_bitop3_b32 (3 vgpr)", k_bitop3},
              {"v_bitop3_b32 (sgpr src)", k_bitop3_s}, {"v_lshl_add_u32", k_lshladd},
              {"v_and_or_b32", k_andor},          {"v_bfe_u32", k_bfe},               {"v_lshlrev_b32 (const shift)", k_lshl}, {"v_lshlrev_b32 (vgpr shift)", k_lshl_v}, {"v_and_b32", k_and},
              {"v_or_b32", k_or}, {"v_xor_b32 (literal)", k_xor_c},
              {"v_mov_b32_dpp quad_perm", k_movdpp}, {"v_xor_b32_dpp quad_perm", k_xordpp}, {"v_fma_f32", k_fma},
              {"v_add_f32", k_addf},              {"v_pk_add_u16", k_pkadd16},        {"v_mov_b32", k_mov},
              {"v_lshlrev_b32_sdwa BYTE_1", k_sdwa}};
    const int iters = 2000;
    // ns = wall clock of the launch / instructions per SIMD (launch overhead included: ~10 us of ~0.1-1 ms)
    for (auto &v : vs) {
        printf("VALU %-28s", v.name);
        for (int wps : {1, 2, 4, 8}) {  // waves per SIMD; 8 = two workgroups of 1024 threads per CU
            double ticks;
            g_grid = wps == 8 ? 512 : 256;
            const int threads = wps == 8 ? 1024 : 256 * wps;
            const double sec = run(v.k, threads, 0, iters, &ticks);
            g_grid = 256;
            printf("  %dw: %.2f tick %.3f ns", wps, ticks / (iters * 64.0 * (wps == 8 ? 4 : wps)) / (wps == 8 ? 2 : 1),
                   sec * 1e9 / (iters * 64.0 * wps));
        }
        printf("  (per wave-instr per SIMD)\n");
    }
    {  // VMEM gathers
        uint32_t *d_tab;
        HIPCHECK(hipMalloc(&d_tab, 65536));
        HIPCHECK(hipMemset(d_tab, 1, 65536));
        for (int threads : {256, 1024}) {
            hipEvent_t e0, e1;
            HIPCHECK(hipEventCreate(&e0));
            HIPCHECK(hipEventCreate(&e1));
            for (int shift : {2, 8}) {
                auto kern = shift == 2 ? k_vmem<2> : k_vmem<8>;
                hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d_tab, d_out, 4, d_ticks);
                HIPCHECK(hipDeviceSynchronize());
                HIPCHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d_tab, d_out, 500, d_ticks);
                HIPCHECK(hipEventRecord(e1));
                HIPCHECK(hipEventSynchronize(e1));
                float ms = 0;
                HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("VMEM global_load_dword gather, %s table, %2d waves/CU: %.3f ns per wave-load per CU\n",
                       shift == 2 ? "1 KiB compact" : "64 KiB replicated", threads / 64, ms * 1e6 / (500.0 * 16 * (threads / 64)));
            }
        }
    }
    // (2) LDS shapes, 16 waves per CU, no VALU
    const size_t L = 65536 + 256;
#define LDSRUN(MODE, NV, PR, LABEL)                                                                             \
    {                                                                                                           \
        double ticks;                                                                                           \
        const double sec = run(k_lds<MODE, NV, PR>, 1024, L, 1000, &ticks);                                     \
        printf("LDS  %-44s nvalu=%2d prio=%d: %.2f tick %.3f ns per wave-load per CU; %.1f tick %.1f ns per iteration (16 waves)\n", \
               LABEL, NV, PR, ticks / (1000.0 * 16 * 16), sec * 1e9 / (1000.0 * 16 * 16), ticks / 1000.0, sec * 1e9 / 1000.0); \
    }
    LDSRUN(0, 0, 0, "ds_read_b32 linear (bank = lane)")
    LDSRUN(1, 0, 0, "ds_read_b32 replicated row (production)")
    LDSRUN(2, 0, 0, "ds_read_b32 replicated row, second half")
    LDSRUN(3, 0, 0, "ds_read_b32 8-byte stride")
    LDSRUN(4, 0, 0, "ds_read_b32 unaligned (+1)")
    LDSRUN(5, 0, 0, "ds_read_u8 replicated row")
    LDSRUN(6, 0, 0, "ds_read_u16 replicated row")
    LDSRUN(7, 0, 0, "ds_read_b64 linear")
    LDSRUN(8, 0, 0, "ds_read_b64 8-byte stride row")
    LDSRUN(9, 0, 0, "ds_read_b32 row of 64 copies (bank = lane)")
    // (3) overlap surface: 16 production look-ups + N independent VALU per iteration
    LDSRUN(1, 8, 0, "mix")
    LDSRUN(1, 16, 0, "mix")
    LDSRUN(1, 24, 0, "mix")
    LDSRUN(1, 32, 0, "mix")
    LDSRUN(1, 40, 0, "mix")
    LDSRUN(1, 48, 0, "mix")
    LDSRUN(1, 16, 1, "mix, waves 4-7,12-15 at prio 1")
    LDSRUN(1, 32, 1, "mix, waves 4-7,12-15 at prio 1")
    LDSRUN(1, 48, 1, "mix, waves 4-7,12-15 at prio 1")
    return 0;
}
