// fused_kernels.hip — the fused schedule: ONE launch per Garble / Eval pass.
//
// A workgroup owns a tile of TI instances and walks ALL levels of the circuit itself; levels are
// separated by a workgroup barrier only (the vector L1 is shared by the waves of a CU, so a
// __syncthreads() makes the labels a level wrote visible to the next one).  Tiles are independent:
// no grid barrier, no inter-workgroup traffic, no per-level launch boundary.  This is what makes
// the narrow levels of real circuits (aes_128: 308 levels, ~119 gates each) fast at batch sizes
// where a level is far too little work for 256 CUs.
//
// Work decomposition inside a level (lanes of the 1024-thread workgroup, in this order):
//   garble: AND  4 lanes per (gate, instance): H(a0), H(a1), H(b0), H(b1)   (circuit/garble.go:362-376)
//           OR   4 lanes: the four enc(a_u, b_v, 0, id)                      (garble.go:421-424)
//           INV  2 lanes: enc(a0), enc(a1)                                   (garble.go:453-454)
//   eval:   AND  2 lanes: H(a, j0), H(b, j1)   OR 1 lane   INV 1 lane       (circuit/eval.go:68-72,93,108)
//   XOR/XNOR 1 lane.
// Every hash lane runs ONE AES through the bank-conflict-free replicated T-table; the 2-4 lanes of a
// gate-instance then exchange their results with DPP quad permutes (register-to-register, no LDS).
// All hash lanes of all gate types share one AES code path, so mixed waves never run it twice.
#include "aes_device.h"
#include "kernels.h"

namespace gc {

constexpr int kFusedThreads = 1024;

// quad_perm DPP controls
constexpr int DPP_XOR1 = 0xB1;   // [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;   // [2,3,0,1]
constexpr int DPP_XOR3 = 0x1B;   // [3,2,1,0]
constexpr int DPP_BC0 = 0x00;    // [0,0,0,0]
constexpr int DPP_BC2 = 0xAA;    // [2,2,2,2]
constexpr int DPP_PAIR0 = 0xA0;  // [0,0,2,2]

template <int CTRL>
__device__ __forceinline__ uint32_t dpp32(uint32_t v) {
    // quad permutes have no invalid source lanes inside a full quad: no "old" value, so no register initialisation
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ uint4 dpp128(uint4 v) {
    return make_uint4(dpp32<CTRL>(v.x), dpp32<CTRL>(v.y), dpp32<CTRL>(v.z), dpp32<CTRL>(v.w));
}

// lane -> (kind, gate, instance, sub-lane) inside one level
struct LanePos {
    int kind;
    uint32_t g, inst, q;
};

// QA / QO / QI: lanes per AND / OR / INV gate-instance (garble 4/4/2, eval 2/1/1 -> log2 2/2/1, 1/0/0)
template <int LQA, int LQO, int LQI>
__device__ __forceinline__ LanePos classify(const Step &st, uint32_t t, uint32_t ti_log2, uint32_t tim) {
    LanePos p{0, 0, 0, 0};
    const uint32_t e_and = (st.n_and << ti_log2) << LQA;
    const uint32_t e_or = e_and + ((st.n_or << ti_log2) << LQO);
    const uint32_t e_inv = e_or + ((st.n_inv << ti_log2) << LQI);
    const uint32_t e_all = e_inv + ((st.count - st.nonfree) << ti_log2);
    if (t < e_and) {
        p.kind = 1;
        p.g = t >> (ti_log2 + LQA);
        p.inst = (t >> LQA) & tim;
        p.q = t & ((1u << LQA) - 1);
    } else if (t < e_or) {
        const uint32_t u = t - e_and;
        p.kind = 2;
        p.g = st.n_and + (u >> (ti_log2 + LQO));
        p.inst = (u >> LQO) & tim;
        p.q = u & ((1u << LQO) - 1);
    } else if (t < e_inv) {
        const uint32_t u = t - e_or;
        p.kind = 3;
        p.g = st.n_and + st.n_or + (u >> (ti_log2 + LQI));
        p.inst = (u >> LQI) & tim;
        p.q = u & ((1u << LQI) - 1);
    } else if (t < e_all) {
        const uint32_t u = t - e_inv;
        p.kind = 4;
        p.g = st.nonfree + (u >> ti_log2);
        p.inst = u & tim;
    }
    return p;
}

template <int LQA, int LQO, int LQI>
__device__ __forceinline__ uint32_t level_lanes(const Step &st, uint32_t ti_log2) {
    return ((st.n_and << ti_log2) << LQA) + ((st.n_or << ti_log2) << LQO) + ((st.n_inv << ti_log2) << LQI) +
           ((st.count - st.nonfree) << ti_log2);
}

enum LaneKind { K_NONE = 0, K_AND = 1, K_OR = 2, K_INV = 3, K_FREE = 4 };

// PROF: per-workgroup s_memtime breakdown (debug builds of the launch only, gc_batch_debug_profile):
//   prof[wg*8 + {0,1,2,3}] = cycles wave 0 spent in {descriptor fetch, label loads, hash+combine+stores,
//   barrier wait}; +4..7 the same for the last wave
#define GC_PROF_MARK(slot)                                             \
    if constexpr (PROF) {                                              \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    \
        const uint64_t now__ = __builtin_amdgcn_s_memtime();           \
        pacc[slot] += now__ - plast;                                   \
        plast = now__;                                                 \
    }

template <int NR, bool PROF>
__global__ __launch_bounds__(kFusedThreads) void k_garble_fused(const GateDesc *__restrict__ descs,
                                                                const Step *__restrict__ steps, uint32_t nsteps,
                                                                uint32_t ninputs, uint32_t ti_log2, size_t w_tile,
                                                                size_t t_tile, uint4 *__restrict__ W,
                                                                const uint4 *__restrict__ Rv, uint4 *__restrict__ T,
                                                                const uint32_t *__restrict__ rk,
                                                                const uint32_t *__restrict__ g_te0,
                                                                uint64_t *__restrict__ prof) {
    __shared__ uint32_t te[kTeDualBytes / 4];
    load_te_dual(te, g_te0);
    uint32_t rkr[4 * (NR + 1)];
    load_round_keys<NR>(rkr, rk);
    __syncthreads();
    const uint32_t lo = te_lane_off();
    const uint32_t TI = 1u << ti_log2, tim = TI - 1;
    uint4 *Wt = W + (size_t)blockIdx.x * w_tile;
    uint4 *Tt = T + (size_t)blockIdx.x * t_tile;
    const uint4 *Rt = Rv + (size_t)blockIdx.x * TI;
    uint64_t pacc[4] = {0, 0, 0, 0}, plast = 0;
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();

    // software pipeline: the descriptor of this thread's first lane of the NEXT level is fetched
    // before the barrier of the current one (it does not depend on anything the level computes)
    Step st_next = steps[0];
    LanePos lp_next = classify<2, 2, 1>(st_next, threadIdx.x, ti_log2, tim);
    GateDesc d_next = lp_next.kind ? descs[st_next.first + lp_next.g] : GateDesc{0, 0, 0, 0};
    for (uint32_t lv = 0; lv < nsteps; lv++) {
        const Step st = st_next;
        const uint32_t e_all = level_lanes<2, 2, 1>(st, ti_log2);
        for (uint32_t t0 = 0; t0 < e_all; t0 += kFusedThreads) {
            const LanePos lp = t0 == 0 ? lp_next : classify<2, 2, 1>(st, t0 + threadIdx.x, ti_log2, tim);
            const int kind = lp.kind;
            const uint32_t g = lp.g, inst = lp.inst, q = lp.q;
            if (kind == K_NONE) continue;
            const GateDesc d = t0 == 0 ? d_next : descs[st.first + g];
            GC_PROF_MARK(0)
            const size_t o_out = ((size_t)(ninputs + st.first + g) << ti_log2) + inst;
            const uint4 va = Wt[((size_t)d.in0 << ti_log2) + inst];
            if (kind == K_FREE) {
                uint4 v = lxor(va, Wt[((size_t)d.in1 << ti_log2) + inst]);
                if ((d.row_op >> kOpShift) == GC_XNOR) v = lxor(v, Rt[inst]);  // garble.go:342-351
                Wt[o_out] = v;
                continue;
            }
            // ---- hash lanes ----
            const uint4 R = Rt[inst];
            uint4 base;         // the L0 label this lane hashes (before the optional ^R)
            uint32_t k[4];
            if (kind == K_OR) {  // e[2u+v] = enc(a_u, b_v, 0, id): K = 2a ^ 4b ^ id  (garble.go:74-83)
                const uint4 vb = Wt[((size_t)d.in1 << ti_log2) + inst];
                const uint4 a = lxor(va, land(R, (q & 2) ? ~0u : 0u));
                const uint4 b = lxor(vb, land(R, (q & 1) ? ~0u : 0u));
                base = make_uint4(a.y, b.y, 0, 0);  // only the S bits are needed afterwards
                make_k(a, b, d.tweak, k);
            } else {  // AND: q = 0..3 -> a0,a1,b0,b1 ; INV: q = 0,1 -> a0,a1.  K = 2x ^ tweak
                const bool second = (kind == K_AND) && (q & 2);
                base = second ? Wt[((size_t)d.in1 << ti_log2) + inst] : va;
                const uint4 x = lxor(base, land(R, (q & 1) ? ~0u : 0u));
                make_k_half(x, d.tweak + (second ? 1u : 0u), k);
            }
            GC_PROF_MARK(1)
            const uint4 h = hash_dual<NR>(k, rkr, te, lo);
            uint4 *row = Tt + ((size_t)(d.row_op & kRowMask) << ti_log2) + inst;

            if (kind == K_AND) {  // garble.go:353-395
                const uint4 p = lxor(h, dpp128<DPP_XOR1>(h));   // lanes 0,1: Ha0^Ha1   lanes 2,3: Hb0^Hb1
                const uint4 a0 = dpp128<DPP_BC0>(base);
                const uint32_t pa = smask(a0);
                const uint32_t pb = (uint32_t)((int32_t)dpp32<DPP_BC2>(base.y) >> 31);
                uint4 v;  // lane 0: WG0, lane 2: WE0
                uint4 tab;
                if (q & 2) {
                    tab = lxor(p, a0);                                     // TE = Hb0^Hb1^a0
                    v = lxor(h, land(lxor(tab, a0), pb));                  // WE0 = Hb0 ^ (pb ? TE^a0 : 0)
                } else {
                    tab = lxor(p, land(R, pb));                            // TG = Ha0^Ha1^(pb?R:0)
                    v = lxor(h, land(tab, pa));                            // WG0 = Ha0 ^ (pa ? TG : 0)
                }
                const uint4 other = dpp128<DPP_XOR2>(v);
                if (q == 0) {
                    Wt[o_out] = lxor(v, other);
                    row[0] = tab;
                } else if (q == 2) {
                    row[TI] = tab;
                }
            } else if (kind == K_INV) {  // garble.go:446-474 (see gc_kernels.hip for the algebra)
                const uint4 p = lxor(h, dpp128<DPP_XOR1>(h));  // E0 ^ E1
                if (q == 0) {
                    const bool s = lbit_s(base);
                    Wt[o_out] = s ? lxor(p, h) : lxor(h, R);  // S(a0) ? E1 : E0^R
                    row[0] = lxor(p, R);
                }
            } else {  // K_OR: garble.go:412-444
                // lane q holds e[q]; pa = S(a_u)^u, pb = S(b_v)^v recover the permute bits of (a0,b0)
                const uint32_t pa = (base.x >> 31) ^ ((q >> 1) & 1), pb = (base.y >> 31) ^ (q & 1);
                const uint32_t l0 = 2 * pa + pb;
                // table[k] = e[k ^ l0]: lane k fetches lane k^l0
                const uint4 x1 = dpp128<DPP_XOR1>(h), x2 = dpp128<DPP_XOR2>(h), x3 = dpp128<DPP_XOR3>(h);
                const uint4 tk = l0 == 0 ? h : l0 == 1 ? x1 : l0 == 2 ? x2 : x3;
                const uint4 t0 = dpp128<DPP_BC0>(tk);
                const uint32_t m0 = l0 == 0 ? ~0u : 0u;
                const uint4 c0 = lxor(t0, land(R, ~m0)), c1 = lxor(t0, land(R, m0));
                if (q == 0) Wt[o_out] = c0;
                else row[(size_t)(q - 1) << ti_log2] = lxor(tk, q == l0 ? c0 : c1);
            }
        }
        if (lv + 1 < nsteps) {  // prefetch the next level's step + this thread's first descriptor
            st_next = steps[lv + 1];
            lp_next = classify<2, 2, 1>(st_next, threadIdx.x, ti_log2, tim);
            if (lp_next.kind) d_next = descs[st_next.first + lp_next.g];
        }
        GC_PROF_MARK(2)
        __syncthreads();
        GC_PROF_MARK(3)
    }
    if constexpr (PROF) {
        if (threadIdx.x == 0 || threadIdx.x == kFusedThreads - 64)
            for (int i = 0; i < 4; i++) prof[(size_t)blockIdx.x * 8 + (threadIdx.x ? 4 : 0) + i] = pacc[i];
    }
}

template <int NR, bool PROF>
__global__ __launch_bounds__(kFusedThreads) void k_eval_fused(const GateDesc *__restrict__ descs,
                                                              const Step *__restrict__ steps, uint32_t nsteps,
                                                              uint32_t ninputs, uint32_t ti_log2, size_t w_tile,
                                                              size_t t_tile, uint4 *__restrict__ W,
                                                              const uint4 *__restrict__ T,
                                                              const uint32_t *__restrict__ rk,
                                                              const uint32_t *__restrict__ g_te0,
                                                              uint64_t *__restrict__ prof) {
    __shared__ uint32_t te[kTeDualBytes / 4];
    load_te_dual(te, g_te0);
    uint32_t rkr[4 * (NR + 1)];
    load_round_keys<NR>(rkr, rk);
    __syncthreads();
    const uint32_t lo = te_lane_off();
    const uint32_t TI = 1u << ti_log2, tim = TI - 1;
    uint4 *Wt = W + (size_t)blockIdx.x * w_tile;
    const uint4 *Tt = T + (size_t)blockIdx.x * t_tile;
    uint64_t pacc[4] = {0, 0, 0, 0}, plast = 0;
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();

    Step st_next = steps[0];
    LanePos lp_next = classify<1, 0, 0>(st_next, threadIdx.x, ti_log2, tim);
    GateDesc d_next = lp_next.kind ? descs[st_next.first + lp_next.g] : GateDesc{0, 0, 0, 0};
    for (uint32_t lv = 0; lv < nsteps; lv++) {
        const Step st = st_next;
        const uint32_t e_all = level_lanes<1, 0, 0>(st, ti_log2);
        for (uint32_t t0 = 0; t0 < e_all; t0 += kFusedThreads) {
            const LanePos lp = t0 == 0 ? lp_next : classify<1, 0, 0>(st, t0 + threadIdx.x, ti_log2, tim);
            const int kind = lp.kind;
            const uint32_t g = lp.g, inst = lp.inst, q = lp.q;
            if (kind == K_NONE) continue;
            const GateDesc d = t0 == 0 ? d_next : descs[st.first + g];
            GC_PROF_MARK(0)
            const size_t o_out = ((size_t)(ninputs + st.first + g) << ti_log2) + inst;
            const uint4 va = Wt[((size_t)d.in0 << ti_log2) + inst];
            if (kind == K_FREE) {  // eval.go:49-51
                Wt[o_out] = lxor(va, Wt[((size_t)d.in1 << ti_log2) + inst]);
                continue;
            }
            const uint4 *row = Tt + ((size_t)(d.row_op & kRowMask) << ti_log2) + inst;
            uint32_t k[4];
            uint4 x = va;  // label hashed by this lane
            uint4 vb = make_uint4(0, 0, 0, 0);
            if (kind == K_AND) {
                if (q) x = Wt[((size_t)d.in1 << ti_log2) + inst];
                make_k_half(x, d.tweak + q, k);
            } else if (kind == K_INV) {
                make_k_half(x, d.tweak, k);
            } else {
                vb = Wt[((size_t)d.in1 << ti_log2) + inst];
                make_k(va, vb, d.tweak, k);
            }
            GC_PROF_MARK(1)
            const uint4 h = hash_dual<NR>(k, rkr, te, lo);
            if (kind == K_AND) {  // eval.go:53-78
                const uint4 tab = row[q ? TI : 0];  // lane 0: TG, lane 1: TE
                const uint4 a = dpp128<DPP_PAIR0>(x);
                uint4 v;
                if (q) v = lxor(h, land(lxor(tab, a), smask(x)));  // WE = H(b) ^ (sb ? TE^a : 0)
                else v = lxor(h, land(tab, smask(x)));             // WG = H(a) ^ (sa ? TG : 0)
                const uint4 other = dpp128<DPP_XOR1>(v);
                if (q == 0) Wt[o_out] = lxor(v, other);
            } else if (kind == K_INV) {  // eval.go:96-109
                Wt[o_out] = lxor(h, land(row[0], smask(x)));
            } else {  // eval.go:80-94
                const uint32_t index = (lbit_s(va) ? 2u : 0u) | (lbit_s(vb) ? 1u : 0u);
                uint4 c = make_uint4(0, 0, 0, 0);
                if (index > 0) c = row[(size_t)(index - 1) << ti_log2];
                Wt[o_out] = lxor(h, c);
            }
        }
        if (lv + 1 < nsteps) {  // prefetch the next level's step + this thread's first descriptor
            st_next = steps[lv + 1];
            lp_next = classify<1, 0, 0>(st_next, threadIdx.x, ti_log2, tim);
            if (lp_next.kind) d_next = descs[st_next.first + lp_next.g];
        }
        GC_PROF_MARK(2)
        __syncthreads();
        GC_PROF_MARK(3)
    }
    if constexpr (PROF) {
        if (threadIdx.x == 0 || threadIdx.x == kFusedThreads - 64)
            for (int i = 0; i < 4; i++) prof[(size_t)blockIdx.x * 8 + (threadIdx.x ? 4 : 0) + i] = pacc[i];
    }
}

void launch_garble_fused(const FusedArgs &a, const BatchGeom &g, hipStream_t s) {
    if (a.nsteps == 0) return;
    dim3 grid(g.ntiles), block(kFusedThreads);
#define GC_GF(NR)                                                                                               \
    if (a.prof)                                                                                                 \
        hipLaunchKernelGGL((k_garble_fused<NR, true>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs,    \
                           g.ti_log2, g.lw.tile_stride, g.lt.tile_stride, a.W, a.R, a.T, a.rk, a.te0, a.prof);  \
    else                                                                                                        \
        hipLaunchKernelGGL((k_garble_fused<NR, false>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs,   \
                           g.ti_log2, g.lw.tile_stride, g.lt.tile_stride, a.W, a.R, a.T, a.rk, a.te0, a.prof)
    switch (a.rounds) {
    case 10: GC_GF(10); break;
    case 12: GC_GF(12); break;
    default: GC_GF(14); break;
    }
#undef GC_GF
}

void launch_eval_fused(const FusedArgs &a, const BatchGeom &g, hipStream_t s) {
    if (a.nsteps == 0) return;
    dim3 grid(g.ntiles), block(kFusedThreads);
#define GC_EF(NR)                                                                                               \
    if (a.prof)                                                                                                 \
        hipLaunchKernelGGL((k_eval_fused<NR, true>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs,      \
                           g.ti_log2, g.lw.tile_stride, g.lt.tile_stride, a.W, (const uint4 *)a.T, a.rk, a.te0, \
                           a.prof);                                                                             \
    else                                                                                                        \
        hipLaunchKernelGGL((k_eval_fused<NR, false>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs,     \
                           g.ti_log2, g.lw.tile_stride, g.lt.tile_stride, a.W, (const uint4 *)a.T, a.rk, a.te0, \
                           a.prof)
    switch (a.rounds) {
    case 10: GC_EF(10); break;
    case 12: GC_EF(12); break;
    default: GC_EF(14); break;
    }
#undef GC_EF
}

}  // namespace gc
