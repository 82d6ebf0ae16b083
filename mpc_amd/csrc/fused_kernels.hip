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
#include "level_gate.h"
#include <cstdlib>

namespace gc {

constexpr int kFusedThreads = 1024;
constexpr int kGroup = 4;  // passes of a level whose loads are issued together (see k_garble_fused)

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

// a label from the wire array; COH: past this CU's L1 (two 64-bit loads of agent scope) — the cooperative one-instance kernels
// read labels that OTHER CUs of the XCD wrote into lines this CU may still hold
template <bool COH>
__device__ __forceinline__ uint4 load_label(const uint4 *p) {
    if constexpr (COH) {
        const uint64_t *q = (const uint64_t *)p;
        const uint64_t a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    } else {
        return *p;
    }
}

template <int NR, bool PROF, int G, bool PRE = true, bool COH = false>
__device__ __forceinline__ void garble_group(const Step &st, const uint32_t t0, const int n_kind, const uint32_t n_g,
                                             const uint32_t n_inst, const uint32_t n_q, const uint32_t n_in0,
                                             const uint32_t n_in1, const uint32_t n_tweak, const uint32_t n_row_op, const GateDesc *__restrict__ descs,
                                             uint32_t ninputs, uint32_t ti_log2, uint32_t tim, uint32_t TI, uint4 *Wt,
                                             uint4 *Tt, const uint4 *Rt, const uint32_t (&rkr)[4 * (NR + 1)],
                                             const uint32_t *te, uint32_t lo, uint64_t (&pacc)[4], uint64_t &plast,
                                             const uint32_t lane_limit = kFusedThreads) {
    LanePos lp[G];
    GateDesc d[G];
    uint4 va[G], vb[G];
#pragma unroll
    for (int p = 0; p < G; p++) {
        if (PRE && t0 == 0 && p == 0) {  // prefetched across the previous level's barrier
            lp[p] = LanePos{n_kind, n_g, n_inst, n_q};
            d[p] = GateDesc{n_in0, n_in1, n_tweak, n_row_op};
        } else {
            // lane_limit (level launches of ONE instance): only the first lanes of the workgroup carry work
            lp[p] = classify<2, 2, 1>(st, threadIdx.x < lane_limit ? t0 + p * kFusedThreads + threadIdx.x : 0xffffffffu, ti_log2, tim);
            d[p] = lp[p].kind ? descs[st.first + lp[p].g] : GateDesc{0, 0, 0, 0};
        }
    }
    GC_PROF_MARK(0)
#pragma unroll
    for (int p = 0; p < G; p++) {
        va[p] = vb[p] = make_uint4(0, 0, 0, 0);
        if (lp[p].kind == K_NONE) continue;
        // AND lanes 2,3 hash operand b; OR and XOR lanes need both operands
        const bool second = lp[p].kind == K_AND && (lp[p].q & 2);
        va[p] = load_label<COH>(Wt + ((size_t)(second ? d[p].in1 : d[p].in0) << ti_log2) + lp[p].inst);
        if (lp[p].kind == K_FREE || lp[p].kind == K_OR) vb[p] = load_label<COH>(Wt + ((size_t)d[p].in1 << ti_log2) + lp[p].inst);
    }
    GC_PROF_MARK(1)
#pragma unroll
    for (int p = 0; p < G; p++) {
        const int kind = lp[p].kind;
        const uint32_t g = lp[p].g, inst = lp[p].inst, q = lp[p].q;
        if (kind == K_NONE) continue;
        const size_t o_out = ((size_t)(ninputs + st.first + g) << ti_log2) + inst;
        if (kind == K_FREE) {
            uint4 v = lxor(va[p], vb[p]);
            if ((d[p].row_op >> kOpShift) == GC_XNOR) v = lxor(v, Rt[inst]);  // garble.go:342-351
            Wt[o_out] = v;
            continue;
        }
        // ---- hash lanes ----
        const uint4 R = Rt[inst];
        uint4 base;  // the L0 label this lane hashes (before the optional ^R)
        uint32_t k[4];
        if (kind == K_OR) {  // e[2u+v] = enc(a_u, b_v, 0, id): K = 2a ^ 4b ^ id  (garble.go:74-83)
            const uint4 a = lxor(va[p], land(R, (q & 2) ? ~0u : 0u));
            const uint4 b = lxor(vb[p], land(R, (q & 1) ? ~0u : 0u));
            base = make_uint4(a.y, b.y, 0, 0);  // only the S bits are needed afterwards
            make_k(a, b, d[p].tweak, k);
        } else {  // AND: q = 0..3 -> a0,a1,b0,b1 ; INV: q = 0,1 -> a0,a1.  K = 2x ^ tweak
            const bool second = (kind == K_AND) && (q & 2);
            base = va[p];
            const uint4 x = lxor(base, land(R, (q & 1) ? ~0u : 0u));
            make_k_half(x, d[p].tweak + (second ? 1u : 0u), k);
        }
        const uint4 h = hash_dual<NR>(k, rkr, te, lo);
        uint4 *row = Tt + ((size_t)(d[p].row_op & kRowMask) << ti_log2) + inst;

        if (kind == K_AND) {  // garble.go:353-395
            const uint4 pp = lxor(h, dpp128<DPP_XOR1>(h));   // lanes 0,1: Ha0^Ha1   lanes 2,3: Hb0^Hb1
            const uint4 a0 = dpp128<DPP_BC0>(base);
            const uint32_t pa = smask(a0);
            const uint32_t pb = (uint32_t)((int32_t)dpp32<DPP_BC2>(base.y) >> 31);
            uint4 v;  // lane 0: WG0, lane 2: WE0
            uint4 tab;
            if (q & 2) {
                tab = lxor(pp, a0);                                    // TE = Hb0^Hb1^a0
                v = lxor(h, land(lxor(tab, a0), pb));                  // WE0 = Hb0 ^ (pb ? TE^a0 : 0)
            } else {
                tab = lxor(pp, land(R, pb));                           // TG = Ha0^Ha1^(pb?R:0)
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
            const uint4 pp = lxor(h, dpp128<DPP_XOR1>(h));  // E0 ^ E1
            if (q == 0) {
                const bool sbit = lbit_s(base);
                Wt[o_out] = sbit ? lxor(pp, h) : lxor(h, R);  // S(a0) ? E1 : E0^R
                row[0] = lxor(pp, R);
            }
        } else {  // K_OR: garble.go:412-444
            // lane q holds e[q]; pa = S(a_u)^u, pb = S(b_v)^v recover the permute bits of (a0,b0)
            const uint32_t pa = (base.x >> 31) ^ ((q >> 1) & 1), pb = (base.y >> 31) ^ (q & 1);
            const uint32_t l0 = 2 * pa + pb;
            // table[k] = e[k ^ l0]: lane k fetches lane k^l0
            const uint4 x1 = dpp128<DPP_XOR1>(h), x2 = dpp128<DPP_XOR2>(h), x3 = dpp128<DPP_XOR3>(h);
            const uint4 tk = l0 == 0 ? h : l0 == 1 ? x1 : l0 == 2 ? x2 : x3;
            const uint4 tz = dpp128<DPP_BC0>(tk);
            const uint32_t m0 = l0 == 0 ? ~0u : 0u;
            const uint4 c0 = lxor(tz, land(R, ~m0)), c1 = lxor(tz, land(R, m0));
            if (q == 0) Wt[o_out] = c0;
            else row[(size_t)(q - 1) << ti_log2] = lxor(tk, q == l0 ? c0 : c1);
        }
    }
    GC_PROF_MARK(2)
}

template <int NR, bool PROF, int G, bool PRE = true, bool COH = false, bool PRE_TAB = false>
__device__ __forceinline__ void eval_group(const Step &st, const uint32_t t0, const int n_kind, const uint32_t n_g,
                                           const uint32_t n_inst, const uint32_t n_q, const uint32_t n_in0,
                                           const uint32_t n_in1, const uint32_t n_tweak, const uint32_t n_row_op, const GateDesc *__restrict__ descs,
                                           uint32_t ninputs, uint32_t ti_log2, uint32_t tim, uint32_t TI, uint4 *Wt,
                                           const uint4 *Tt, const uint32_t (&rkr)[4 * (NR + 1)], const uint32_t *te,
                                           uint32_t lo, uint64_t (&pacc)[4], uint64_t &plast,
                                           const uint32_t lane_limit = kFusedThreads, const uint4 n_tab = uint4{0, 0, 0, 0}) {
    LanePos lp[G];
    GateDesc d[G];
    uint4 va[G], vb[G], tab[G];
#pragma unroll
    for (int p = 0; p < G; p++) {
        if (PRE && t0 == 0 && p == 0) {  // prefetched across the previous level's barrier
            lp[p] = LanePos{n_kind, n_g, n_inst, n_q};
            d[p] = GateDesc{n_in0, n_in1, n_tweak, n_row_op};
        } else {
            lp[p] = classify<1, 0, 0>(st, threadIdx.x < lane_limit ? t0 + p * kFusedThreads + threadIdx.x : 0xffffffffu, ti_log2, tim);
            d[p] = lp[p].kind ? descs[st.first + lp[p].g] : GateDesc{0, 0, 0, 0};
        }
    }
    GC_PROF_MARK(0)
#pragma unroll
    for (int p = 0; p < G; p++) {
        va[p] = vb[p] = tab[p] = make_uint4(0, 0, 0, 0);
        const int kind = lp[p].kind;
        if (kind == K_NONE) continue;
        const uint32_t inst = lp[p].inst, q = lp[p].q;
        // AND lane 1 hashes operand b; OR and XOR lanes need both operands
        va[p] = load_label<COH>(Wt + ((size_t)((kind == K_AND && q) ? d[p].in1 : d[p].in0) << ti_log2) + inst);
        if (kind == K_FREE || kind == K_OR) vb[p] = load_label<COH>(Wt + ((size_t)d[p].in1 << ti_log2) + inst);
        const uint4 *row = Tt + ((size_t)(d[p].row_op & kRowMask) << ti_log2) + inst;
        if (PRE_TAB && PRE && t0 == 0 && p == 0) tab[p] = n_tab;  // fetched with the descriptor, before the level's barrier
        else if (kind == K_AND) tab[p] = row[q ? TI : 0];  // lane 0: TG, lane 1: TE
        else if (kind == K_INV) tab[p] = row[0];
    }
    GC_PROF_MARK(1)
#pragma unroll
    for (int p = 0; p < G; p++) {
        const int kind = lp[p].kind;
        const uint32_t g = lp[p].g, inst = lp[p].inst, q = lp[p].q;
        if (kind == K_NONE) continue;
        const size_t o_out = ((size_t)(ninputs + st.first + g) << ti_log2) + inst;
        if (kind == K_FREE) {  // eval.go:49-51
            Wt[o_out] = lxor(va[p], vb[p]);
            continue;
        }
        uint32_t k[4];
        const uint4 x = va[p];  // label hashed by this lane
        if (kind == K_AND) make_k_half(x, d[p].tweak + q, k);
        else if (kind == K_INV) make_k_half(x, d[p].tweak, k);
        else make_k(va[p], vb[p], d[p].tweak, k);
        const uint4 h = hash_dual<NR>(k, rkr, te, lo);
        if (kind == K_AND) {  // eval.go:53-78
            const uint4 a = dpp128<DPP_PAIR0>(x);
            uint4 v;
            if (q) v = lxor(h, land(lxor(tab[p], a), smask(x)));  // WE = H(b) ^ (sb ? TE^a : 0)
            else v = lxor(h, land(tab[p], smask(x)));             // WG = H(a) ^ (sa ? TG : 0)
            const uint4 other = dpp128<DPP_XOR1>(v);
            if (q == 0) Wt[o_out] = lxor(v, other);
        } else if (kind == K_INV) {  // eval.go:96-109
            Wt[o_out] = lxor(h, land(tab[p], smask(x)));
        } else {  // eval.go:80-94: the row index depends on the labels themselves
            const uint32_t index = (lbit_s(va[p]) ? 2u : 0u) | (lbit_s(vb[p]) ? 1u : 0u);
            uint4 c = make_uint4(0, 0, 0, 0);
            if (index > 0) c = (Tt + ((size_t)(d[p].row_op & kRowMask) << ti_log2) + inst)[(size_t)(index - 1) << ti_log2];
            Wt[o_out] = lxor(h, c);
        }
    }
    GC_PROF_MARK(2)
}


// ---- narrow levels, column-sliced: four lanes per AES block -------------------------------------------------------------
// A level of ONE pass is bound by one wave's instruction issue (a lone wave issues one VALU per 2.5 ns whatever the opcode,
// DESIGN §4): ~520 instructions for the pass around the hash plus 14 x 48 for a wide-form AES = 3.0 us per level on the
// synthetic W = 64 rows (profiles/r03_exp_hbm_wire_overlap.txt).  When the level has no OR gate and its lanes still fit one
// pass with FOUR lanes per block, it runs column-sliced like the narrow units of the flat kernels (aes_device.h:
// hash_col_whitened; fused_flat_kernels.hip: garble_hash_narrow): one 32-bit state column per lane, ~15 instructions per
// round, the hash lanes spread over four times the waves — SIMDs that were idle.  Same arithmetic as the wide form: label
// word W_c (big-endian column c) sits at dword c ^ 1; the q ^ 1 / q ^ 2 partners of a gate are 4 / 8 lanes away inside the
// row of 16.  Lane space of such a level: [4 x wide hash lanes | free lanes], J = threadIdx.x.
constexpr uint32_t kColKeyTab = kTeDualBytes;  // LDS byte address of the column-addressable round keys (behind the table)

template <int NR>
__device__ __forceinline__ void load_col_keys(uint32_t *te, const uint32_t *__restrict__ rk) {
    if (threadIdx.x < 4 * (NR + 1)) {  // the last round key folded with the first (the hashes run on whitened blocks)
        uint32_t kv = rk[threadIdx.x];
        if (threadIdx.x >= 4 * NR) kv ^= rk[threadIdx.x - 4 * NR];
        te[kColKeyTab / 4 + threadIdx.x] = kv;
    }
}
__device__ __forceinline__ uint32_t col_pair4(uint32_t v) {  // value of the lane 4 further on (q even) / 4 back (q odd)
    uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0x5, false);
    return (uint32_t)__builtin_amdgcn_update_dpp((int)r, (int)v, 0x114, 0xf, 0xa, false);
}
__device__ __forceinline__ uint32_t col_whiten(uint32_t xc, uint32_t xc1, uint32_t c, uint32_t tweak, uint32_t k0) {
    const uint32_t kcol = __builtin_amdgcn_alignbit(xc, c == 3 ? 0u : xc1, 31);
    return xor3(kcol, c == 3 ? tweak : 0u, k0);
}
__device__ __forceinline__ uint32_t label_word(const uint4 *label, uint32_t byte_off) {
    return *(const uint32_t *)((const char *)label + byte_off);
}
// lanes of a level in column form (LQA / LQI: log2 of the wide lanes per AND / INV gate-instance); 0xffffffff: has OR gates
template <int LQA, int LQI>
__device__ __forceinline__ uint32_t col_lanes(const Step &st, uint32_t ti_log2) {
    if (st.n_or) return 0xffffffffu;
    return ((((st.n_and << ti_log2) << LQA) + ((st.n_inv << ti_log2) << LQI)) << 2) + ((st.count - st.nonfree) << ti_log2);
}
// this lane's place in the column form and its gate's descriptor (fetched before the previous level's barrier)
struct ColPos {
    uint32_t kind, g, inst, q, c;  // kind: K_AND / K_INV / K_FREE / K_NONE
};
template <int LQA, int LQI>
__device__ __forceinline__ ColPos col_classify(const Step &st, uint32_t J, uint32_t ti_log2, uint32_t tim) {
    ColPos p{K_NONE, 0, 0, 0, 0};
    const uint32_t e_and = ((st.n_and << ti_log2) << LQA) << 2, ncol = e_and + (((st.n_inv << ti_log2) << LQI) << 2);
    if (J < e_and) {
        const uint32_t w = J >> 2;
        p.kind = K_AND, p.c = J & 3u, p.q = w & ((1u << LQA) - 1), p.inst = (w >> LQA) & tim, p.g = w >> (ti_log2 + LQA);
    } else if (J < ncol) {
        const uint32_t w = (J - e_and) >> 2;
        p.kind = K_INV, p.c = J & 3u, p.q = w & ((1u << LQI) - 1), p.inst = (w >> LQI) & tim, p.g = st.n_and + (w >> (ti_log2 + LQI));
    } else if (J - ncol < ((st.count - st.nonfree) << ti_log2)) {
        const uint32_t u = J - ncol;
        p.kind = K_FREE, p.inst = u & tim, p.g = st.nonfree + (u >> ti_log2);
    }
    return p;
}

template <int NR>
__device__ __forceinline__ void garble_col_pass(const Step &st, const ColPos &p, const GateDesc &d, uint32_t ninputs,
                                                uint32_t ti_log2, uint32_t TI, uint4 *Wt, uint4 *Tt, const uint4 *Rt, uint32_t lo) {
    if (p.kind == K_NONE) return;
    const uint32_t inst = p.inst;
    uint4 *outl = Wt + ((size_t)(ninputs + st.first + p.g) << ti_log2) + inst;
    const uint4 *la = Wt + ((size_t)d.in0 << ti_log2) + inst;
    if (p.kind == K_FREE) {
        uint4 v = lxor(*la, Wt[((size_t)d.in1 << ti_log2) + inst]);
        if ((d.row_op >> kOpShift) == GC_XNOR) v = lxor(v, Rt[inst]);  // garble.go:342-351
        *outl = v;
        return;
    }
    const uint32_t c = p.c, q = p.q, wo = (c ^ 1u) << 2, wo1 = (((c + 1u) ^ 1u) << 2) & 12u;
    const uint4 *lb = p.kind == K_AND ? Wt + ((size_t)d.in1 << ti_log2) + inst : la;
    const uint4 *lown = (q & 2u) ? lb : la;  // INV lanes have q < 2
    const uint32_t keyaddr = kColKeyTab + (c << 2);
    const uint32_t bc = label_word(lown, wo), bc1 = label_word(lown, wo1);
    const uint32_t a0c = label_word(la, wo), a0y = label_word(la, 4), b0y = label_word(lb, 4);
    const uint32_t rc = label_word(Rt + inst, wo), rc1 = label_word(Rt + inst, wo1);
    const uint32_t k0 = *(lds_u32 *)(uintptr_t)keyaddr;
    const uint32_t modd = (q & 1u) ? ~0u : 0u;
    const uint32_t xc = __builtin_amdgcn_bitop3_b32(bc, rc, modd, 0x78), xc1 = __builtin_amdgcn_bitop3_b32(bc1, rc1, modd, 0x78);
    const uint32_t h = hash_col_whitened<NR>(col_whiten(xc, xc1, c, d.tweak + (q >> 1), k0), keyaddr, lo);
    char *row0 = (char *)(Tt + ((size_t)(d.row_op & kRowMask) << ti_log2) + inst) + wo;
    const uint32_t pp = h ^ col_pair4(h);
    if (p.kind == K_AND) {  // garble.go:353-395, one column
        const uint32_t m2 = (q & 2u) ? ~0u : 0u;
        const uint32_t pa = (uint32_t)((int32_t)a0y >> 31), pb = (uint32_t)((int32_t)b0y >> 31);
        const uint32_t mk = m2 ? pb : pa, rm = pb & ~m2;
        const uint32_t w = __builtin_amdgcn_bitop3_b32(pp, rc, rm, 0x78);
        const uint32_t tab = __builtin_amdgcn_bitop3_b32(w, a0c, m2, 0x78);
        const uint32_t v = __builtin_amdgcn_bitop3_b32(h, w, mk, 0x78);
        if (!(q & 1u)) *(uint32_t *)(row0 + (((q & 2u) ? (size_t)TI : 0) << 4)) = tab;
        uint32_t o = v ^ (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, true);  // the q ^ 2 partner: row_ror:8
        asm volatile("" : "+v"(o));
        if (q == 0) *(uint32_t *)((char *)outl + wo) = o;
    } else if (q == 0) {  // INV, garble.go:446-474
        *(uint32_t *)row0 = pp ^ rc;
        *(uint32_t *)((char *)outl + wo) = h ^ (((int32_t)a0y < 0) ? pp : rc);
    }
}

template <int NR>
__device__ __forceinline__ void eval_col_pass(const Step &st, const ColPos &p, const GateDesc &d, uint32_t ninputs,
                                              uint32_t ti_log2, uint32_t TI, uint4 *Wt, const uint4 *Tt, uint32_t lo) {
    if (p.kind == K_NONE) return;
    const uint32_t inst = p.inst;
    uint4 *outl = Wt + ((size_t)(ninputs + st.first + p.g) << ti_log2) + inst;
    const uint4 *la = Wt + ((size_t)d.in0 << ti_log2) + inst;
    if (p.kind == K_FREE) {  // eval.go:49-51
        *outl = lxor(*la, Wt[((size_t)d.in1 << ti_log2) + inst]);
        return;
    }
    const uint32_t c = p.c, q = p.q, wo = (c ^ 1u) << 2, wo1 = (((c + 1u) ^ 1u) << 2) & 12u;
    const uint4 *lb = p.kind == K_AND ? Wt + ((size_t)d.in1 << ti_log2) + inst : la;
    const uint4 *lown = q ? lb : la;  // AND lane 1 hashes operand b (INV: q = 0)
    const uint32_t keyaddr = kColKeyTab + (c << 2);
    const uint32_t tab = label_word(Tt + ((size_t)(d.row_op & kRowMask) << ti_log2) + inst + (q ? TI : 0u), wo);
    const uint32_t xc = label_word(lown, wo), xc1 = label_word(lown, wo1), xy = label_word(lown, 4);
    const uint32_t ac = label_word(la, wo), k0 = *(lds_u32 *)(uintptr_t)keyaddr;
    const uint32_t h = hash_col_whitened<NR>(col_whiten(xc, xc1, c, d.tweak + q, k0), keyaddr, lo);
    const uint32_t sm = (uint32_t)((int32_t)xy >> 31);
    if (p.kind == K_AND) {  // eval.go:53-78: lane 0 WG = H(a) ^ (sa ? TG : 0), lane 1 WE = H(b) ^ (sb ? TE ^ a : 0)
        const uint32_t v = __builtin_amdgcn_bitop3_b32(h, __builtin_amdgcn_bitop3_b32(tab, ac, q ? ~0u : 0u, 0x78), sm, 0x78);
        uint32_t o = v ^ (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x104, 0xf, 0xf, true);  // + the lane 4 further on (q = 1)
        asm volatile("" : "+v"(o));
        if (q == 0) *(uint32_t *)((char *)outl + wo) = o;
    } else {  // eval.go:96-109
        *(uint32_t *)((char *)outl + wo) = __builtin_amdgcn_bitop3_b32(h, tab, sm, 0x78);
    }
}

// circuits whose EVERY level runs column-sliced in one pass — two for the odd wider level — (no OR gate anywhere, 4 x hash
// lanes + free lanes <= 2 048 at this tile size; the host decides: FusedArgs::narrow_col): a loop of [this lane's gate, fetched before the barrier | pass | barrier]
template <int NR>
__global__ __launch_bounds__(kFusedThreads) void k_garble_col(const GateDesc *__restrict__ descs, const Step *__restrict__ steps,
                                                              uint32_t nsteps, uint32_t ninputs, uint32_t ti_log2, size_t w_tile,
                                                              size_t t_tile, uint4 *__restrict__ W, const uint4 *__restrict__ Rv,
                                                              uint4 *__restrict__ T, const uint32_t *__restrict__ rk,
                                                              const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeDualBytes / 4 + 64];  // + the column-addressable round keys (kColKeyTab)
    __shared__ uint4 Rt[64];
    load_te_dual(te, g_te0);
    load_col_keys<NR>(te, rk);
    const uint32_t TI = 1u << ti_log2, tim = TI - 1;
    if (threadIdx.x < TI) Rt[threadIdx.x] = Rv[(size_t)blockIdx.x * TI + threadIdx.x];
    __syncthreads();
    const uint32_t lo = te_lane_off();
    uint4 *Wt = W + (size_t)blockIdx.x * w_tile;
    uint4 *Tt = T + (size_t)blockIdx.x * t_tile;
    Step st_next = steps[0];
    ColPos cp_next = col_classify<2, 1>(st_next, threadIdx.x, ti_log2, tim);
    GateDesc d_next = descs[st_next.first + cp_next.g];
    for (uint32_t lv = 0; lv < nsteps; lv++) {
        const Step st = st_next;
        garble_col_pass<NR>(st, cp_next, d_next, ninputs, ti_log2, TI, Wt, Tt, Rt, lo);
        for (uint32_t J = kFusedThreads; J < col_lanes<2, 1>(st, ti_log2); J += kFusedThreads) {  // a level of a second pass
            const ColPos cp = col_classify<2, 1>(st, J + threadIdx.x, ti_log2, tim);
            garble_col_pass<NR>(st, cp, descs[st.first + cp.g], ninputs, ti_log2, TI, Wt, Tt, Rt, lo);
        }
        if (lv + 1 < nsteps) {
            st_next = steps[lv + 1];
            cp_next = col_classify<2, 1>(st_next, threadIdx.x, ti_log2, tim);
            d_next = descs[st_next.first + cp_next.g];
        }
        __syncthreads();
    }
}

template <int NR>
__global__ __launch_bounds__(kFusedThreads) void k_eval_col(const GateDesc *__restrict__ descs, const Step *__restrict__ steps,
                                                            uint32_t nsteps, uint32_t ninputs, uint32_t ti_log2, size_t w_tile,
                                                            size_t t_tile, uint4 *__restrict__ W, const uint4 *__restrict__ T,
                                                            const uint32_t *__restrict__ rk, const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeDualBytes / 4 + 64];
    load_te_dual(te, g_te0);
    load_col_keys<NR>(te, rk);
    __syncthreads();
    const uint32_t TI = 1u << ti_log2, tim = TI - 1;
    const uint32_t lo = te_lane_off();
    uint4 *Wt = W + (size_t)blockIdx.x * w_tile;
    const uint4 *Tt = T + (size_t)blockIdx.x * t_tile;
    Step st_next = steps[0];
    ColPos cp_next = col_classify<1, 0>(st_next, threadIdx.x, ti_log2, tim);
    GateDesc d_next = descs[st_next.first + cp_next.g];
    for (uint32_t lv = 0; lv < nsteps; lv++) {
        const Step st = st_next;
        eval_col_pass<NR>(st, cp_next, d_next, ninputs, ti_log2, TI, Wt, Tt, lo);
        for (uint32_t J = kFusedThreads; J < col_lanes<1, 0>(st, ti_log2); J += kFusedThreads) {
            const ColPos cp = col_classify<1, 0>(st, J + threadIdx.x, ti_log2, tim);
            eval_col_pass<NR>(st, cp, descs[st.first + cp.g], ninputs, ti_log2, TI, Wt, Tt, lo);
        }
        if (lv + 1 < nsteps) {
            st_next = steps[lv + 1];
            cp_next = col_classify<1, 0>(st_next, threadIdx.x, ti_log2, tim);
            d_next = descs[st_next.first + cp_next.g];
        }
        __syncthreads();
    }
}

// ---- wide levels: hash waves and free waves ----------------------------------------------------------------------------
// A level of several passes used to run as load -> wait -> hash -> store per group of four passes with all sixteen waves in
// step, and the label traffic and the AES time of a level ADDED UP (profiles/r03_exp_hbm_wire_overlap.txt: the cost of the
// traffic is the time the waves spend issuing it into a memory pipe that takes ~19 GB/s per CU of these 64-byte pieces, and a
// wave that is stuck on a vector-memory instruction does not hash; pipelining the passes of ALL waves removed every wait for
// data and not a microsecond).  So the two kinds of work go to different waves of the workgroup:
//   hash waves (the first H)  walk the level's AND / OR / INV lanes, 64 H per pass, as a software pipeline of depth one —
//                             operand loads of pass p+1 and the descriptor of p+2 in flight under the AES of pass p, the only
//                             vmcnt(0) at the END of the hash (what it waits for is one AES old), stores behind it — their
//                             few vector-memory instructions per pass never stall them for long;
//   free waves (the other 16 - H) stream the XOR / XNOR lanes in groups of four passes, bound by the memory pipe alone.
// Both finish at the level's barrier.  The two operand sets of the pipeline alternate (A, B): copying a just-loaded register
// is a use, and a use in front of the hashes is a wait in front of the hashes; for the same reason every load of the pipeline
// is unconditional (idle lanes re-read the level's first descriptor and its operand) and nothing is zero-filled.
// lane position packed into one register (the pipeline keeps five of these alive): kind | q << 3 | inst << 8
__device__ __forceinline__ uint32_t pos_pack(const LanePos &lp) { return (uint32_t)lp.kind | (lp.q << 3) | (lp.inst << 8); }
__device__ __forceinline__ int pos_kind(uint32_t pos) { return (int)(pos & 7u); }
__device__ __forceinline__ uint32_t pos_q(uint32_t pos) { return (pos >> 3) & 3u; }
__device__ __forceinline__ uint32_t pos_inst(uint32_t pos) { return pos >> 8; }

struct PassDesc {
    uint32_t pos, g;
    GateDesc d;
};
struct PassIn {
    uint32_t pos, g, tweak, row_op;
    uint4 va, vb, tab;  // tab: evaluator only
};
struct PassOut {
    uint4 w, t;
    uint32_t w_idx, t_idx;  // index into the tile's wire / table array, ~0u: nothing to store
};

__device__ __forceinline__ void garble_issue_labels(PassIn &o, const PassDesc &pd, uint32_t ti_log2, const uint4 *Wt) {
    o.pos = pd.pos;
    o.g = pd.g;
    o.tweak = pd.d.tweak;
    o.row_op = pd.d.row_op;
    const int kind = pos_kind(pd.pos);
    // (fields first, selection after: a select between two struct fields becomes a select between two addresses and
    // sends the whole descriptor set to scratch)
    const uint32_t in0 = pd.d.in0, in1 = pd.d.in1, inst = pos_inst(pd.pos);
    const bool second = kind == K_AND && (pos_q(pd.pos) & 2);
    const uint32_t ia = second ? in1 : in0;
    // va of every lane (idle lanes: operand 0 of the level's first gate), vb where it is used and left undefined elsewhere —
    // no zero-fill: a merge of {0, loaded} is again a copy that waits
    o.va = Wt[((size_t)ia << ti_log2) + inst];
    if (kind == K_FREE || kind == K_OR) o.vb = Wt[((size_t)in1 << ti_log2) + inst];
}

__device__ __forceinline__ void eval_issue_labels(PassIn &o, const PassDesc &pd, uint32_t ti_log2, uint32_t TI, const uint4 *Wt,
                                                  const uint4 *Tt) {
    o.pos = pd.pos;
    o.g = pd.g;
    o.tweak = pd.d.tweak;
    o.row_op = pd.d.row_op;
    const int kind = pos_kind(pd.pos);
    const uint32_t inst = pos_inst(pd.pos), q = pos_q(pd.pos), in0 = pd.d.in0, in1 = pd.d.in1;
    const uint32_t ia = (kind == K_AND && q) ? in1 : in0;
    const uint32_t row = ((pd.d.row_op & kRowMask) << ti_log2) + inst + ((kind == K_AND && q) ? TI : 0u);
    o.va = Wt[((size_t)ia << ti_log2) + inst];
    if (kind == K_FREE || kind == K_OR) o.vb = Wt[((size_t)in1 << ti_log2) + inst];
    if (kind == K_AND || kind == K_INV) o.tab = Tt[row];  // AND lane 0: TG, lane 1: TE
}

// one pass of the garbler: the same algebra as garble_group, results returned instead of stored
template <int NR>
__device__ __forceinline__ PassOut garble_pass(const Step &st, const PassIn &in, uint32_t ninputs, uint32_t ti_log2, uint32_t TI,
                                               const uint4 *Rt, const uint32_t (&rkr)[4 * (NR + 1)], const uint32_t *te,
                                               uint32_t lo) {
    PassOut o;
    o.w = o.t = make_uint4(0, 0, 0, 0);
    o.w_idx = o.t_idx = 0xffffffffu;
    const int kind = pos_kind(in.pos);
    const uint32_t g = in.g, inst = pos_inst(in.pos), q = pos_q(in.pos);
    if (kind == K_NONE) return o;
    const uint32_t o_out = ((ninputs + st.first + g) << ti_log2) + inst;
    if (kind == K_FREE) {
        uint4 v = lxor(in.va, in.vb);
        if ((in.row_op >> kOpShift) == GC_XNOR) v = lxor(v, Rt[inst]);  // garble.go:342-351
        o.w = v;
        o.w_idx = o_out;
        return o;
    }
    const uint4 R = Rt[inst];
    uint4 base;
    uint32_t k[4];
    if (kind == K_OR) {  // garble.go:74-83
        const uint4 a = lxor(in.va, land(R, (q & 2) ? ~0u : 0u));
        const uint4 b = lxor(in.vb, land(R, (q & 1) ? ~0u : 0u));
        base = make_uint4(a.y, b.y, 0, 0);
        make_k(a, b, in.tweak, k);
    } else {
        const bool second = (kind == K_AND) && (q & 2);
        base = in.va;
        const uint4 x = lxor(base, land(R, (q & 1) ? ~0u : 0u));
        make_k_half(x, in.tweak + (second ? 1u : 0u), k);
    }
    const uint4 h = hash_dual<NR>(k, rkr, te, lo);
    const uint32_t row = ((in.row_op & kRowMask) << ti_log2) + inst;
    if (kind == K_AND) {  // garble.go:353-395
        const uint4 pp = lxor(h, dpp128<DPP_XOR1>(h));
        const uint4 a0 = dpp128<DPP_BC0>(base);
        const uint32_t pa = smask(a0);
        const uint32_t pb = (uint32_t)((int32_t)dpp32<DPP_BC2>(base.y) >> 31);
        uint4 v, tab;
        if (q & 2) {
            tab = lxor(pp, a0);
            v = lxor(h, land(lxor(tab, a0), pb));
        } else {
            tab = lxor(pp, land(R, pb));
            v = lxor(h, land(tab, pa));
        }
        const uint4 other = dpp128<DPP_XOR2>(v);
        if (q == 0) {
            o.w = lxor(v, other);
            o.w_idx = o_out;
            o.t = tab;
            o.t_idx = row;
        } else if (q == 2) {
            o.t = tab;
            o.t_idx = row + TI;
        }
    } else if (kind == K_INV) {  // garble.go:446-474
        const uint4 pp = lxor(h, dpp128<DPP_XOR1>(h));
        if (q == 0) {
            const bool sbit = lbit_s(base);
            o.w = sbit ? lxor(pp, h) : lxor(h, R);
            o.w_idx = o_out;
            o.t = lxor(pp, R);
            o.t_idx = row;
        }
    } else {  // K_OR: garble.go:412-444
        const uint32_t pa = (base.x >> 31) ^ ((q >> 1) & 1), pb = (base.y >> 31) ^ (q & 1);
        const uint32_t l0 = 2 * pa + pb;
        const uint4 x1 = dpp128<DPP_XOR1>(h), x2 = dpp128<DPP_XOR2>(h), x3 = dpp128<DPP_XOR3>(h);
        const uint4 tk = l0 == 0 ? h : l0 == 1 ? x1 : l0 == 2 ? x2 : x3;
        const uint4 tz = dpp128<DPP_BC0>(tk);
        const uint32_t m0 = l0 == 0 ? ~0u : 0u;
        const uint4 c0 = lxor(tz, land(R, ~m0)), c1 = lxor(tz, land(R, m0));
        if (q == 0) {
            o.w = c0;
            o.w_idx = o_out;
        } else {
            o.t = lxor(tk, q == l0 ? c0 : c1);
            o.t_idx = row + ((q - 1) << ti_log2);
        }
    }
    return o;
}

// one pass of the evaluator (eval_group's algebra)
template <int NR>
__device__ __forceinline__ PassOut eval_pass(const Step &st, const PassIn &in, uint32_t ninputs, uint32_t ti_log2, const uint4 *Tt,
                                             const uint32_t (&rkr)[4 * (NR + 1)], const uint32_t *te, uint32_t lo) {
    PassOut o;
    o.w = o.t = make_uint4(0, 0, 0, 0);
    o.w_idx = o.t_idx = 0xffffffffu;
    const int kind = pos_kind(in.pos);
    const uint32_t g = in.g, inst = pos_inst(in.pos), q = pos_q(in.pos);
    if (kind == K_NONE) return o;
    const uint32_t o_out = ((ninputs + st.first + g) << ti_log2) + inst;
    if (kind == K_FREE) {  // eval.go:49-51
        o.w = lxor(in.va, in.vb);
        o.w_idx = o_out;
        return o;
    }
    uint32_t k[4];
    const uint4 x = in.va;
    if (kind == K_AND) make_k_half(x, in.tweak + q, k);
    else if (kind == K_INV) make_k_half(x, in.tweak, k);
    else make_k(in.va, in.vb, in.tweak, k);
    const uint4 h = hash_dual<NR>(k, rkr, te, lo);
    if (kind == K_AND) {  // eval.go:53-78
        const uint4 a = dpp128<DPP_PAIR0>(x);
        uint4 v;
        if (q) v = lxor(h, land(lxor(in.tab, a), smask(x)));
        else v = lxor(h, land(in.tab, smask(x)));
        const uint4 other = dpp128<DPP_XOR1>(v);
        if (q == 0) {
            o.w = lxor(v, other);
            o.w_idx = o_out;
        }
    } else if (kind == K_INV) {  // eval.go:96-109
        o.w = lxor(h, land(in.tab, smask(x)));
        o.w_idx = o_out;
    } else {  // eval.go:80-94
        const uint32_t index = (lbit_s(in.va) ? 2u : 0u) | (lbit_s(in.vb) ? 1u : 0u);
        uint4 c = make_uint4(0, 0, 0, 0);
        if (index > 0) c = (Tt + ((size_t)(in.row_op & kRowMask) << ti_log2) + inst)[(size_t)(index - 1) << ti_log2];
        o.w = lxor(h, c);
        o.w_idx = o_out;
    }
    return o;
}

// one group of the pipeline: `cur` holds the operands of group g (loads issued one group earlier), `nxt` receives those of
// group g+1.  The level loops call this with two operand sets in alternation, so no register is ever copied between them —
// a copy of a just-loaded register is a use, and a use in front of the hashes is a wait in front of the hashes.
template <int LQA, int LQO, int LQI>
__device__ __forceinline__ void role_fetch_desc(PassDesc &o, const Step &st, uint32_t t, uint32_t e_hash,
                                                const GateDesc *__restrict__ descs, uint32_t ti_log2, uint32_t tim) {
    const LanePos lp = classify<LQA, LQO, LQI>(st, t < e_hash ? t : 0xffffffffu, ti_log2, tim);
    o.pos = pos_pack(lp);
    o.g = lp.g;
    o.d = descs[st.first + lp.g];
}

// one pass of the hash waves: cur holds the operands of this pass, nxt receives those of the next one
template <int NR>
__device__ __forceinline__ void garble_hash_step(const Step &st, uint32_t t_next2, uint32_t e_hash, PassDesc &dn, PassIn &cur,
                                                 PassIn &nxt, const GateDesc *__restrict__ descs, uint32_t ninputs,
                                                 uint32_t ti_log2, uint32_t tim, uint32_t TI, uint4 *Wt, uint4 *Tt,
                                                 const uint4 *Rt, const uint32_t (&rkr)[4 * (NR + 1)], const uint32_t *te,
                                                 uint32_t lo) {
    garble_issue_labels(nxt, dn, ti_log2, Wt);
    role_fetch_desc<2, 2, 1>(dn, st, t_next2, e_hash, descs, ti_log2, tim);
    const PassOut o = garble_pass<NR>(st, cur, ninputs, ti_log2, TI, Rt, rkr, te, lo);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loads above and the stores of the previous pass, one AES old
    if (o.w_idx != 0xffffffffu) Wt[o.w_idx] = o.w;
    if (o.t_idx != 0xffffffffu) Tt[o.t_idx] = o.t;
}

template <int NR>
__device__ __forceinline__ void garble_hash_role(const Step &st, uint32_t e_hash, uint32_t H, const LanePos &lp0,
                                                 const GateDesc &d0, const PassDesc &pre1, const GateDesc *__restrict__ descs,
                                                 uint32_t ninputs, uint32_t ti_log2, uint32_t tim, uint32_t TI, uint4 *Wt,
                                                 uint4 *Tt, const uint4 *Rt, const uint32_t (&rkr)[4 * (NR + 1)],
                                                 const uint32_t *te, uint32_t lo) {
    const uint32_t LP = H << 6, NP = (e_hash + LP - 1) / LP;
    // the descriptors of the first two passes were fetched before the barrier of the previous level (pass 0 of a hash wave
    // is the lane numbering of a whole-workgroup pass: lp0 / d0 are the kernel's prefetch for single-pass levels)
    PassDesc dn;
    PassIn A, B;
    dn.pos = threadIdx.x < e_hash ? pos_pack(lp0) : 0u;
    dn.g = threadIdx.x < e_hash ? lp0.g : 0u;
    dn.d = d0;
    garble_issue_labels(A, dn, ti_log2, Wt);
    dn = pre1;
    for (uint32_t p = 0; p < NP; p += 2) {
        garble_hash_step<NR>(st, (p + 2) * LP + threadIdx.x, e_hash, dn, A, B, descs, ninputs, ti_log2, tim, TI, Wt, Tt, Rt, rkr, te, lo);
        if (p + 1 < NP)
            garble_hash_step<NR>(st, (p + 3) * LP + threadIdx.x, e_hash, dn, B, A, descs, ninputs, ti_log2, tim, TI, Wt, Tt, Rt, rkr, te, lo);
    }
}

template <int NR>
__device__ __forceinline__ void eval_hash_step(const Step &st, uint32_t t_next2, uint32_t e_hash, PassDesc &dn, PassIn &cur,
                                               PassIn &nxt, const GateDesc *__restrict__ descs, uint32_t ninputs,
                                               uint32_t ti_log2, uint32_t tim, uint32_t TI, uint4 *Wt, const uint4 *Tt,
                                               const uint32_t (&rkr)[4 * (NR + 1)], const uint32_t *te, uint32_t lo) {
    eval_issue_labels(nxt, dn, ti_log2, TI, Wt, Tt);
    role_fetch_desc<1, 0, 0>(dn, st, t_next2, e_hash, descs, ti_log2, tim);
    const PassOut o = eval_pass<NR>(st, cur, ninputs, ti_log2, Tt, rkr, te, lo);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (o.w_idx != 0xffffffffu) Wt[o.w_idx] = o.w;
}

template <int NR>
__device__ __forceinline__ void eval_hash_role(const Step &st, uint32_t e_hash, uint32_t H, const LanePos &lp0,
                                               const GateDesc &d0, const PassDesc &pre1, const GateDesc *__restrict__ descs,
                                               uint32_t ninputs, uint32_t ti_log2, uint32_t tim, uint32_t TI, uint4 *Wt,
                                               const uint4 *Tt, const uint32_t (&rkr)[4 * (NR + 1)], const uint32_t *te,
                                               uint32_t lo) {
    const uint32_t LP = H << 6, NP = (e_hash + LP - 1) / LP;
    PassDesc dn;
    PassIn A, B;
    dn.pos = threadIdx.x < e_hash ? pos_pack(lp0) : 0u;
    dn.g = threadIdx.x < e_hash ? lp0.g : 0u;
    dn.d = d0;
    eval_issue_labels(A, dn, ti_log2, TI, Wt, Tt);
    dn = pre1;
    for (uint32_t p = 0; p < NP; p += 2) {
        eval_hash_step<NR>(st, (p + 2) * LP + threadIdx.x, e_hash, dn, A, B, descs, ninputs, ti_log2, tim, TI, Wt, Tt, rkr, te, lo);
        if (p + 1 < NP)
            eval_hash_step<NR>(st, (p + 3) * LP + threadIdx.x, e_hash, dn, B, A, descs, ninputs, ti_log2, tim, TI, Wt, Tt, rkr, te, lo);
    }
}

// the free waves: XOR / XNOR lanes (the same for both sides but for the offset of an XNOR, garble.go:342-351 / eval.go:49-51),
// groups of kGroup passes of 64 F lanes
template <bool GARBLE>
__device__ __forceinline__ void free_role(const Step &st, uint32_t n_free_lanes, uint32_t first_lane, uint32_t F,
                                          const GateDesc *__restrict__ descs, uint32_t ninputs, uint32_t ti_log2, uint32_t tim,
                                          uint4 *Wt, const uint4 *Rt) {
    const uint32_t LP = F << 6, fl = threadIdx.x - first_lane;
    for (uint32_t u0 = 0; u0 < n_free_lanes; u0 += kGroup * LP) {
        uint32_t gate[kGroup], inst[kGroup];
        bool on[kGroup];
        GateDesc d[kGroup];
        uint4 va[kGroup], vb[kGroup];
#pragma unroll
        for (int p = 0; p < kGroup; p++) {
            const uint32_t u = u0 + p * LP + fl;
            on[p] = u < n_free_lanes;
            gate[p] = on[p] ? st.nonfree + (u >> ti_log2) : 0u;
            inst[p] = u & tim;
            d[p] = descs[st.first + gate[p]];
        }
#pragma unroll
        for (int p = 0; p < kGroup; p++) {
            va[p] = Wt[((size_t)d[p].in0 << ti_log2) + inst[p]];
            vb[p] = Wt[((size_t)d[p].in1 << ti_log2) + inst[p]];
        }
#pragma unroll
        for (int p = 0; p < kGroup; p++) {
            uint4 v = lxor(va[p], vb[p]);
            if (GARBLE && (d[p].row_op >> kOpShift) == GC_XNOR) v = lxor(v, Rt[inst[p]]);
            if (on[p]) Wt[((size_t)(ninputs + st.first + gate[p]) << ti_log2) + inst[p]] = v;
        }
    }
}

// hash waves H of a level with e_hash hash lanes and n_free free lanes (the other F = 16 - H waves stream the free lanes):
// the waves are shared out in proportion to the work, a hash lane counting 3.4 free lanes — the ratio at which both sides
// of the synthetic sweep have their optimum in steady state (W = 1 024 / 16 384, f = 0.17, H = 7 .. 14 measured after ten
// warm-up passes: garbler, 2 784 hash lanes against 3 400 free ones, best with F = 4; evaluator, 1 392 against 3 400, with
// F = 7; f = 0.5: F = 1 - 2; profiles/r03_exp_hbm_wire_overlap.txt).  No free gates -> 16, no hashed gates -> 0, never more
// hash waves than the level has lanes for.  tune != 0: H fixed, for measurements.
__device__ __forceinline__ uint32_t role_hash_waves(uint32_t e_hash, uint32_t n_free, uint32_t tune) {
    if (n_free == 0) return 16;
    if (e_hash == 0) return 0;
    const uint32_t need = (e_hash + 63) >> 6;
    if (tune) return need < tune ? need : tune;
    const uint32_t den = 10u * n_free + 34u * e_hash;
    uint32_t F = (160u * n_free + (den >> 1)) / den;
    F = F < 1 ? 1 : F > 8 ? 8 : F;
    const uint32_t H = 16 - F;
    return need < H ? need : H;
}

template <int NR, bool PROF>
__global__ __launch_bounds__(kFusedThreads) void k_garble_fused(const GateDesc *__restrict__ descs,
                                                                const Step *__restrict__ steps, uint32_t nsteps,
                                                                uint32_t ninputs, uint32_t ti_log2, size_t w_tile,
                                                                size_t t_tile, uint4 *__restrict__ W,
                                                                const uint4 *__restrict__ Rv, uint4 *__restrict__ T,
                                                                const uint32_t *__restrict__ rk,
                                                                const uint32_t *__restrict__ g_te0,
                                                                uint64_t *__restrict__ prof, uint32_t tune) {
    __shared__ uint32_t te[kTeDualBytes / 4];
    load_te_dual(te, g_te0);
    uint32_t rkr[4 * (NR + 1)];
    load_round_keys<NR>(rkr, rk);
    __syncthreads();
    const uint32_t lo = te_lane_off();
    const uint32_t TI = 1u << ti_log2, tim = TI - 1;
    uint4 *Wt = W + (size_t)blockIdx.x * w_tile;
    uint4 *Tt = T + (size_t)blockIdx.x * t_tile;
    // the tile's offsets R live in LDS: a hash lane reads its R on the way into the AES, and a global load there would make
    // that point wait for every vector load in flight (vmcnt), the prefetched operands of the next pass included
    __shared__ uint4 Rt[64];
    if (threadIdx.x < TI) Rt[threadIdx.x] = Rv[(size_t)blockIdx.x * TI + threadIdx.x];
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t pacc[4] = {0, 0, 0, 0}, plast = 0;
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();

    // Passes of a level (1024 lanes each) run in GROUPS of kGroup: first the descriptors of all passes of the group, then
    // all their operand labels, then hash / combine / stores pass by pass.  On gfx9 loads and stores share vmcnt and
    // complete out of order with each other, so the first load after a store waits for EVERY store before it
    // (vmcnt(0)): with one pass at a time that drain (1-2 us) was paid per pass — 60-80 % of the kernel on wide levels
    // (scripts/prof_fused_synth.py).  Now it is paid once per group, overlapped with the next group's loads.
    // the next level's step record and the descriptors of its first group are fetched before the barrier of the current
    // level (they do not depend on anything the level computes): narrow, deep circuits pay one load latency per level
    // instead of three.  Only the single-pass instantiation takes the prefetched descriptor: in the grouped one the
    // choice between the two descriptor sources (prefetched registers / a load) is what this compiler has turned into
    // faulting code under small perturbations of the surrounding loop (the instrumented build on mixed levels, every
    // variant of the loop tried in round 2), and a level of several passes does not notice one more load latency.
    Step st_next = steps[0];
    LanePos lp_next = classify<2, 2, 1>(st_next, threadIdx.x, ti_log2, tim);
    GateDesc d_next = lp_next.kind ? descs[st_next.first + lp_next.g] : GateDesc{0, 0, 0, 0};
    // wide levels (hash waves / free waves): the split of the next level and the hash waves' second descriptor
    uint32_t e_hash_next = level_lanes<2, 2, 1>(st_next, ti_log2) - ((st_next.count - st_next.nonfree) << ti_log2);
    uint32_t H_next = level_lanes<2, 2, 1>(st_next, ti_log2) > (uint32_t)kFusedThreads ? role_hash_waves(e_hash_next, level_lanes<2, 2, 1>(st_next, ti_log2) - e_hash_next, tune) : 0u;
    PassDesc pre1;
    if (!PROF && wave < H_next) role_fetch_desc<2, 2, 1>(pre1, st_next, (H_next << 6) + threadIdx.x, e_hash_next, descs, ti_log2, tim);
    for (uint32_t lv = 0; lv < nsteps; lv++) {
        const Step st = st_next;
        const uint32_t e_all = level_lanes<2, 2, 1>(st, ti_log2);
        const uint32_t e_hash = e_hash_next, H = H_next;
        // a level of one pass (narrow, deep circuits) takes the single-pass instantiation: no classification or
        // zero-fill of empty passes on its critical path
        if (e_all <= (uint32_t)kFusedThreads) garble_group<NR, PROF, 1>(st, 0u, lp_next.kind, lp_next.g, lp_next.inst, lp_next.q, d_next.in0, d_next.in1, d_next.tweak, d_next.row_op, descs, ninputs, ti_log2, tim, TI, Wt, Tt, Rt, rkr, te, lo, pacc, plast);
        else if (!PROF) {
            if (wave < H) garble_hash_role<NR>(st, e_hash, H, lp_next, d_next, pre1, descs, ninputs, ti_log2, tim, TI, Wt, Tt, Rt, rkr, te, lo);
            else free_role<true>(st, e_all - e_hash, H << 6, 16 - H, descs, ninputs, ti_log2, tim, Wt, Rt);
        } else
            // (instrumented builds) the groups of a level in opposite orders for the two halves of the workgroup's waves (quartets 0, 2 forwards;
            // 1, 3 backwards): hash lanes fill the first passes of a level and free lanes the last ones, so while one half
            // hashes the other half moves labels (+4 % on the wide synthetic rows; groups of a level are independent)
            for (uint32_t tt = 0, ng = (e_all + kGroup * kFusedThreads - 1) / (kGroup * kFusedThreads); tt < ng; tt++)
                garble_group<NR, PROF, kGroup, false>(st, (((threadIdx.x >> 8) & 1) ? ng - 1 - tt : tt) * (kGroup * kFusedThreads), lp_next.kind, lp_next.g, lp_next.inst, lp_next.q, d_next.in0, d_next.in1, d_next.tweak, d_next.row_op, descs, ninputs, ti_log2, tim, TI, Wt, Tt, Rt, rkr, te, lo, pacc, plast);
        if (lv + 1 < nsteps) {
            st_next = steps[lv + 1];
            lp_next = classify<2, 2, 1>(st_next, threadIdx.x, ti_log2, tim);
            if (lp_next.kind) d_next = descs[st_next.first + lp_next.g];
            e_hash_next = level_lanes<2, 2, 1>(st_next, ti_log2) - ((st_next.count - st_next.nonfree) << ti_log2);
            H_next = level_lanes<2, 2, 1>(st_next, ti_log2) > (uint32_t)kFusedThreads ? role_hash_waves(e_hash_next, level_lanes<2, 2, 1>(st_next, ti_log2) - e_hash_next, tune) : 0u;
            if (!PROF && wave < H_next) role_fetch_desc<2, 2, 1>(pre1, st_next, (H_next << 6) + threadIdx.x, e_hash_next, descs, ti_log2, tim);
        }
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
                                                              uint64_t *__restrict__ prof, uint32_t tune) {
    __shared__ uint32_t te[kTeDualBytes / 4];
    load_te_dual(te, g_te0);
    uint32_t rkr[4 * (NR + 1)];
    load_round_keys<NR>(rkr, rk);
    __syncthreads();
    const uint32_t lo = te_lane_off();
    const uint32_t TI = 1u << ti_log2, tim = TI - 1;
    uint4 *Wt = W + (size_t)blockIdx.x * w_tile;
    const uint4 *Tt = T + (size_t)blockIdx.x * t_tile;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t pacc[4] = {0, 0, 0, 0}, plast = 0;
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();

    // groups of kGroup passes: descriptors, then operand labels and table rows, then hashes and stores (see the garbler)
    // the next level's step record and the descriptors of its first group are fetched before the barrier of the current
    // level (they do not depend on anything the level computes): narrow, deep circuits pay one load latency per level
    // instead of three
    Step st_next = steps[0];
    LanePos lp_next = classify<1, 0, 0>(st_next, threadIdx.x, ti_log2, tim);
    GateDesc d_next = lp_next.kind ? descs[st_next.first + lp_next.g] : GateDesc{0, 0, 0, 0};
    // wide levels (hash waves / free waves): the split of the next level and the hash waves' second descriptor
    uint32_t e_hash_next = level_lanes<1, 0, 0>(st_next, ti_log2) - ((st_next.count - st_next.nonfree) << ti_log2);
    uint32_t H_next = level_lanes<1, 0, 0>(st_next, ti_log2) > (uint32_t)kFusedThreads ? role_hash_waves(e_hash_next, level_lanes<1, 0, 0>(st_next, ti_log2) - e_hash_next, tune) : 0u;
    PassDesc pre1;
    if (!PROF && wave < H_next) role_fetch_desc<1, 0, 0>(pre1, st_next, (H_next << 6) + threadIdx.x, e_hash_next, descs, ti_log2, tim);
    for (uint32_t lv = 0; lv < nsteps; lv++) {
        const Step st = st_next;
        const uint32_t e_all = level_lanes<1, 0, 0>(st, ti_log2);
        const uint32_t e_hash = e_hash_next, H = H_next;
        // a level of one pass (narrow, deep circuits) takes the single-pass instantiation: no classification or
        // zero-fill of empty passes on its critical path
        if (e_all <= (uint32_t)kFusedThreads) eval_group<NR, PROF, 1>(st, 0u, lp_next.kind, lp_next.g, lp_next.inst, lp_next.q, d_next.in0, d_next.in1, d_next.tweak, d_next.row_op, descs, ninputs, ti_log2, tim, TI, Wt, Tt, rkr, te, lo, pacc, plast);
        else if (!PROF) {
            if (wave < H) eval_hash_role<NR>(st, e_hash, H, lp_next, d_next, pre1, descs, ninputs, ti_log2, tim, TI, Wt, Tt, rkr, te, lo);
            else free_role<false>(st, e_all - e_hash, H << 6, 16 - H, descs, ninputs, ti_log2, tim, Wt, nullptr);
        } else
            for (uint32_t tt = 0, ng = (e_all + kGroup * kFusedThreads - 1) / (kGroup * kFusedThreads); tt < ng; tt++)
                eval_group<NR, PROF, kGroup, false>(st, (((threadIdx.x >> 8) & 1) ? ng - 1 - tt : tt) * (kGroup * kFusedThreads), lp_next.kind, lp_next.g, lp_next.inst, lp_next.q, d_next.in0, d_next.in1, d_next.tweak, d_next.row_op, descs, ninputs, ti_log2, tim, TI, Wt, Tt, rkr, te, lo, pacc, plast);
        if (lv + 1 < nsteps) {
            st_next = steps[lv + 1];
            lp_next = classify<1, 0, 0>(st_next, threadIdx.x, ti_log2, tim);
            if (lp_next.kind) d_next = descs[st_next.first + lp_next.g];
            e_hash_next = level_lanes<1, 0, 0>(st_next, ti_log2) - ((st_next.count - st_next.nonfree) << ti_log2);
            H_next = level_lanes<1, 0, 0>(st_next, ti_log2) > (uint32_t)kFusedThreads ? role_hash_waves(e_hash_next, level_lanes<1, 0, 0>(st_next, ti_log2) - e_hash_next, tune) : 0u;
            if (!PROF && wave < H_next) role_fetch_desc<1, 0, 0>(pre1, st_next, (H_next << 6) + threadIdx.x, e_hash_next, descs, ti_log2, tim);
        }
        __syncthreads();
        GC_PROF_MARK(3)
    }
    if constexpr (PROF) {
        if (threadIdx.x == 0 || threadIdx.x == kFusedThreads - 64)
            for (int i = 0; i < 4; i++) prof[(size_t)blockIdx.x * 8 + (threadIdx.x ? 4 : 0) + i] = pacc[i];
    }
}

// ---- ONE instance, one launch per level, lanes along the gates -------------------------------------------------------
// Lanes of a level per workgroup.  A full workgroup of 1 024 hash lanes is 1 024 AES blocks on ONE CU: 4.4 us at the
// core's 16-wave rate, while 250 CUs idle (a level of a 131 072-gate step has ~3 600 lanes).  256 lanes per workgroup
// (4 waves, one per SIMD; the other 12 waves only help to build the table) spread the level over four times as many CUs:
// the blocks of a workgroup take 2.2 us.
constexpr uint32_t kLevelLanes = 64;
// A single instance on the fused kernels is one workgroup on one CU: a streamed SSA-step circuit of 131 072 gates spends
// 0.78 ms there, AES-bound on that CU.  When a level has several passes of work, the level's lanes are spread over
// workgroups instead — pass k of the level is workgroup k — and the levels become launches (the kernel boundary is
// the inter-workgroup barrier; the launches are recorded once in a hipGraph per circuit and replayed).  Same lane
// decomposition and the same code as a pass of the fused kernels (garble_group / eval_group with one pass, TI = 1).
template <int NR>
__global__ __launch_bounds__(kFusedThreads) void k_garble_level1(const GateDesc *__restrict__ descs, Step st, uint32_t ninputs,
                                                                 uint4 *__restrict__ W, const uint4 *__restrict__ Rv,
                                                                 uint4 *__restrict__ T, const uint32_t *__restrict__ rk,
                                                                 const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeDualBytes / 4];
    uint32_t rkr[4 * (NR + 1)];
    if (blockIdx.x * kLevelLanes < ((st.n_and + st.n_or) << 2) + (st.n_inv << 1)) {  // this pass has hash lanes
        load_te_dual(te, g_te0);
        load_round_keys<NR>(rkr, rk);
        __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < 4 * (NR + 1); i++) rkr[i] = 0;
    }
    uint64_t pacc[4] = {0, 0, 0, 0}, plast = 0;
    garble_group<NR, false, 1, false>(st, blockIdx.x * kLevelLanes, 0, 0u, 0u, 0u, 0u, 0u, 0u, 0u, descs, ninputs, 0u, 0u, 1u, W,
                                      T, Rv, rkr, te, te_lane_off(), pacc, plast, kLevelLanes);
}

template <int NR>
__global__ __launch_bounds__(kFusedThreads) void k_eval_level1(const GateDesc *__restrict__ descs, Step st, uint32_t ninputs,
                                                               uint4 *__restrict__ W, const uint4 *__restrict__ T,
                                                               const uint32_t *__restrict__ rk,
                                                               const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeDualBytes / 4];
    uint32_t rkr[4 * (NR + 1)];
    if (blockIdx.x * kLevelLanes < (st.n_and << 1) + st.n_or + st.n_inv) {
        load_te_dual(te, g_te0);
        load_round_keys<NR>(rkr, rk);
        __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < 4 * (NR + 1); i++) rkr[i] = 0;
    }
    uint64_t pacc[4] = {0, 0, 0, 0}, plast = 0;
    eval_group<NR, false, 1, false>(st, blockIdx.x * kLevelLanes, 0, 0u, 0u, 0u, 0u, 0u, 0u, 0u, descs, ninputs, 0u, 0u, 1u, W, T,
                                    rkr, te, te_lane_off(), pacc, plast, kLevelLanes);
}

// ---- ONE instance, ONE launch: the levels behind a barrier in L2 ------------------------------------------------------------
// A level launch costs 5.9 us for ~3 000 lanes of work (profiles/r03_stream_big_kernel_stats.csv): 2 - 2.5 us from one
// kernel to the next one that depends on it, the 64 KiB table of every hashing workgroup built again, two dependent round
// trips for descriptor and operand.  Here kCoopGroups workgroups of 256 threads stay resident for the whole pass (table
// and round keys once) and meet at a counter between levels.  They are the workgroups 0, 8, 16, ... of the grid: workgroups
// go round-robin over the 8 XCDs, so these sit on ONE XCD and share its L2 — a label is visible to the other CUs as soon as
// its store is acknowledged (the vector L1 writes through), the counter lives in that L2, and the readers fetch labels
// past their L1; no write-back or invalidate of L2, which an agent-scope release / acquire would cost at every level.
// The placement is checked, not assumed: gc_ctx runs k_coop_selftest once (XCC_ID of every group, values handed round
// through the barrier) and keeps the level launches if it fails; every wait is bounded (kCoopPolls), a workgroup that
// gives up raises ctl->error and the pass is reported as failed.
constexpr uint32_t kCoopThreads = 256;
// The bound of a wait is a number of POLLS, not a span of time (until round 6: 40 M s_memtime ticks, ~36 ms): a queue that the
// hardware scheduler takes off the GPU for a while — a second process with a context on the same GPU is enough: BENCH_r05's C
// host failed so once in ~10 runs at config 5's size, "lost a workgroup and could not be repeated" — stops every workgroup of
// the pass together, the clock runs on, and a time bound then fires in ALL of them (and in the stand-by, whose ~2 s backstop was
// what made the failure fatal).  Polls only happen while the wave runs.  One poll is a load past the L1 (0.3 - 0.8 us): 2^16
// polls are 20 - 50 ms of RUNNING time, what a workgroup that is really missing costs once.
constexpr uint32_t kCoopPolls = 1u << 16;
// the stand-by's backstop, in sleeps of 100 x 64 clocks (3 - 6 us each): 2^21 of them are ~10 s of running time
constexpr uint32_t kCoopStandbySleeps = 1u << 21;

// barrier number `gen` (1, 2, ... within one launch): one atomic add, then a plain load past the L1 until every group has
// added.  Measured on one XCD: an atomic add takes 0.7 us to come back whatever its scope; 32 workgroups that poll the
// counter with read-modify-writes queue at its L2 bank (5 - 6 us per barrier); per-group flags (a store, one load of all 32
// flags per poll, no atomic at all) cost 2.3 us; this costs 1.9 us.  The error flag / the clock every 64th poll.
// (in two halves, so that what does not depend on the level — the next level's descriptor — is fetched while the add travels)
__device__ __forceinline__ void coop_arrive(CoopCtl *ctl) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's stores are in L2
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&ctl->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void coop_wait(CoopCtl *ctl, uint32_t gen) {
    if (threadIdx.x == 0) {
        const uint32_t target = gen * kCoopGroups;
        uint32_t spins = 0;
        while (__hip_atomic_load(&ctl->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if ((++spins & 63u) == 0) {
                if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (spins > kCoopPolls) {
                    // (the host hears of it from the stand-by: 1 once the pass has been done again, 2 if that was not possible)
                    __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
    // no cache invalidate: buffer_inv sc1 walks the L2 (12 us measured); the labels are read past the L1 instead
    // (load_label<true>), everything else the pass reads is constant
}

// end of a pass: the last workgroup to get here puts the counters back to zero for the next pass on the stream (every
// group is past its last wait: thread 0 is the one that waits)
__device__ __forceinline__ void coop_leave(CoopCtl *ctl) {
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(&ctl->left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kCoopGroups - 1) {
        __hip_atomic_store(&ctl->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->left, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl->passes, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// testing (GC_COOP_FORCE_TIMEOUT): in pass number ctl->drop_at one workgroup takes no part — to the others it looks exactly
// like a workgroup that did not become resident in time
__device__ __forceinline__ bool coop_dropped(CoopCtl *ctl, uint32_t rank) {
    if (rank != 5) return false;
    const uint32_t at = __hip_atomic_load(&ctl->drop_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (at == 0 || __hip_atomic_load(&ctl->passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 != at) return false;
    coop_leave(ctl);
    return true;
}

// The stand-by of a pass (kernels.h: launch_coop): workgroup number kCoopGroups of the same XCD.  te: the kernel's LDS table
// area (the classic tables take its first 4 KiB).
template <int NR, bool EVAL>
__device__ __noinline__ void coop_standby(CoopCtl *ctl, uint32_t seq, const GateDesc *__restrict__ descs, const Step *__restrict__ steps,
                                          uint32_t nsteps, uint32_t ninputs, uint4 *__restrict__ W, const uint4 *__restrict__ Rv,
                                          uint4 *__restrict__ T, const uint32_t *__restrict__ rk, const uint32_t *__restrict__ g_te0,
                                          const StoreXchg *xp, uint32_t *te) {
    __shared__ uint32_t ended;
    if (threadIdx.x == 0) {
        // the last workgroup to leave counts the pass (ctl->passes becomes seq + 1); every wait inside the pass is bounded, so
        // this one ends.  The bound here is a backstop (~10 s of running time): a pass that has NOT ended by then — or a count the
        // host and the device do not agree on — cannot be repaired from here: the host is told that the results are not to be used (2).
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        uint32_t ok = 1, sleeps = 0;
        while ((int32_t)(__hip_atomic_load(&ctl->passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) <= 0) {
            __builtin_amdgcn_s_sleep(100);
            if (++sleeps > kCoopStandbySleeps) {  // (counted, not timed: see kCoopPolls)
                ok = 0;
                break;
            }
        }
        if (!ok)
            if (uint32_t *host_err = ctl->host_err) {
                // what the stand-by saw when it gave up (words 1.. of the pinned block: gc_ctx_coop_check puts them into the
                // error text): its pass number, the device's count of ended passes, arrivals, leavers, the error flag, ticks waited
                host_err[1] = seq;
                host_err[2] = __hip_atomic_load(&ctl->passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                host_err[3] = __hip_atomic_load(&ctl->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                host_err[4] = __hip_atomic_load(&ctl->left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                host_err[5] = __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                host_err[6] = (uint32_t)((__builtin_amdgcn_s_memtime() - t0) >> 10);
                host_err[7] = nsteps;
                __hip_atomic_store(host_err, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        ended = ok;
    }
    __syncthreads();
    if (!ended) return;
    if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;  // the rule: a good pass
    load_te_tables(te, g_te0);
    if (xp)  // Get through in[] again: the failed pass left the wire store alone (it skips its scatter once the flag is up)
        for (uint32_t i = threadIdx.x; i < ninputs; i += kCoopThreads) W[i] = xp->store[xp->in_idx[i]];
    __syncthreads();
    uint4 R = make_uint4(0, 0, 0, 0);
    if (!EVAL) R = Rv[0];
    for (uint32_t lv = 0; lv < nsteps; lv++) {
        const Step st = steps[lv];
        for (uint32_t g = threadIdx.x; g < st.count; g += kCoopThreads) {
            const GateDesc d = descs[st.first + g];
            uint4 *out = W + ninputs + st.first + g;  // desc k writes slot ninputs + k; one instance: column 0, stride 1
            if (EVAL) eval_one<NR>(d, 0, 1, W, (const uint4 *)T, out, rk, te);
            else garble_one<NR>(d, 0, 1, W, R, T, out, rk, te);
        }
        __syncthreads();  // (one workgroup, one CU: the level's labels are in its L1 / the XCD's L2 for the next level)
    }
    if (xp)
        for (uint32_t j = threadIdx.x; j < xp->nout; j += kCoopThreads)
            if (xp->out_idx[j] != 0xffffffffu) xp->store[xp->out_idx[j]] = W[xp->out_slots[j]];
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ctl->repaired, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->error, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the host: a pass lost a workgroup AND has been done again (never overwrites a 2)
        if (uint32_t *host_err = ctl->host_err) __hip_atomic_fetch_max(host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__device__ __forceinline__ void coop_barrier(CoopCtl *ctl, uint32_t gen) {
    coop_arrive(ctl);
    coop_wait(ctl, gen);
}

// the input labels of the pass from the wire store, then a barrier (generation 1)
__device__ __forceinline__ void coop_gather(const StoreXchg *xp, uint4 *W, uint32_t ninputs, uint32_t rank, CoopCtl *ctl) {
    if (!xp) return;
    const uint4 *store = xp->store;
    const uint32_t *in_idx = xp->in_idx;
    for (uint32_t i = rank * kCoopThreads + threadIdx.x; i < ninputs; i += kCoopGroups * kCoopThreads) W[i] = store[in_idx[i]];
    coop_arrive(ctl);
    coop_wait(ctl, 1);
}
// behind the last level: a barrier (generation gen), then the output labels into the wire store
__device__ __forceinline__ void coop_scatter(const StoreXchg *xp, const uint4 *W, uint32_t rank, CoopCtl *ctl, uint32_t gen) {
    if (!xp) return;
    const StoreXchg x = *xp;
    if (x.nout == 0) return;
    coop_arrive(ctl);
    coop_wait(ctl, gen);
    // a pass that lost a workgroup leaves the wire store alone: k_coop_repair does the pass again behind it, from the same inputs
    if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    for (uint32_t j = rank * kCoopThreads + threadIdx.x; j < x.nout; j += kCoopGroups * kCoopThreads) {
        const uint32_t idx = x.out_idx[j];
        if (idx != 0xffffffffu) x.store[idx] = load_label<true>(W + x.out_slots[j]);
    }
}

template <int NR>
__global__ __launch_bounds__(kCoopThreads) void k_garble_coop(const GateDesc *__restrict__ descs, const Step *__restrict__ steps,
                                                              uint32_t nsteps, uint32_t ninputs, uint4 *__restrict__ W,
                                                              const uint4 *__restrict__ Rv, uint4 *__restrict__ T,
                                                              const uint32_t *__restrict__ rk,
                                                              const uint32_t *__restrict__ g_te0, CoopCtl *ctl,
                                                              const StoreXchg *xp, uint32_t seq) {
    if (blockIdx.x & 7) return;
    const uint32_t rank = blockIdx.x >> 3;
    __shared__ uint32_t te[kTeDualBytes / 4];
    if (rank == kCoopGroups) {
        coop_standby<NR, false>(ctl, seq, descs, steps, nsteps, ninputs, W, Rv, T, rk, g_te0, xp, te);
        return;
    }
    if (coop_dropped(ctl, rank)) return;
    load_te_dual(te, g_te0);
    uint32_t rkr[4 * (NR + 1)];
    load_round_keys<NR>(rkr, rk);
    __syncthreads();
    const uint32_t lo = te_lane_off();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t pacc[4] = {0, 0, 0, 0}, plast = 0;
    // wave-sized chunks of a level's lanes, dealt round over the groups first: the hash lanes (the first ones of a level)
    // spread over all CUs.  The descriptor of this wave's first chunk of the NEXT level is fetched before the barrier.
    const uint32_t c0 = wave * kCoopGroups + rank, stride = (kCoopThreads / 64) * kCoopGroups;
    const uint32_t t_first = (c0 << 6) + (threadIdx.x & 63u);
    Step st_next = steps[0];
    LanePos lp_next = classify<2, 2, 1>(st_next, t_first, 0u, 0u);
    GateDesc d_next = lp_next.kind ? descs[st_next.first + lp_next.g] : GateDesc{0, 0, 0, 0};
    coop_gather(xp, W, ninputs, rank, ctl);
    for (uint32_t lv = 0; lv < nsteps; lv++) {
        const Step st = st_next;
        const uint32_t chunks = (level_lanes<2, 2, 1>(st, 0u) + 63u) >> 6;
        if (c0 < chunks)
            garble_group<NR, false, 1, true, true>(st, 0u, lp_next.kind, lp_next.g, lp_next.inst, lp_next.q, d_next.in0, d_next.in1,
                                                   d_next.tweak, d_next.row_op, descs, ninputs, 0u, 0u, 1u, W, T, Rv, rkr, te, lo,
                                                   pacc, plast, kCoopThreads);
        for (uint32_t c = c0 + stride; c < chunks; c += stride)
            garble_group<NR, false, 1, false, true>(st, (c - wave) << 6, 0, 0u, 0u, 0u, 0u, 0u, 0u, 0u, descs, ninputs, 0u, 0u, 1u, W,
                                                    T, Rv, rkr, te, lo, pacc, plast, kCoopThreads);
        if (lv + 1 < nsteps) {
            coop_arrive(ctl);
            st_next = steps[lv + 1];
            lp_next = classify<2, 2, 1>(st_next, t_first, 0u, 0u);
            if (lp_next.kind) d_next = descs[st_next.first + lp_next.g];
            coop_wait(ctl, lv + 1 + (xp ? 1u : 0u));
        }
    }
    coop_scatter(xp, W, rank, ctl, nsteps + 1);
    coop_leave(ctl);
}

// the table row an evaluator lane of ONE instance needs (TG for AND lane 0, TE for lane 1, the INV row)
__device__ __forceinline__ uint4 eval_tab_row(const LanePos &lp, const GateDesc &d, const uint4 *T) {
    if (lp.kind != K_AND && lp.kind != K_INV) return make_uint4(0, 0, 0, 0);
    return T[(size_t)(d.row_op & kRowMask) + ((lp.kind == K_AND && lp.q) ? 1u : 0u)];
}

template <int NR>
__global__ __launch_bounds__(kCoopThreads) void k_eval_coop(const GateDesc *__restrict__ descs, const Step *__restrict__ steps,
                                                            uint32_t nsteps, uint32_t ninputs, uint4 *__restrict__ W,
                                                            const uint4 *__restrict__ T, const uint32_t *__restrict__ rk,
                                                            const uint32_t *__restrict__ g_te0, CoopCtl *ctl,
                                                            const StoreXchg *xp, uint32_t seq) {
    if (blockIdx.x & 7) return;
    const uint32_t rank = blockIdx.x >> 3;
    __shared__ uint32_t te[kTeDualBytes / 4];
    if (rank == kCoopGroups) {
        coop_standby<NR, true>(ctl, seq, descs, steps, nsteps, ninputs, W, nullptr, const_cast<uint4 *>(T), rk, g_te0, xp, te);
        return;
    }
    if (coop_dropped(ctl, rank)) return;
    load_te_dual(te, g_te0);
    uint32_t rkr[4 * (NR + 1)];
    load_round_keys<NR>(rkr, rk);
    __syncthreads();
    const uint32_t lo = te_lane_off();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t pacc[4] = {0, 0, 0, 0}, plast = 0;
    const uint32_t c0 = wave * kCoopGroups + rank, stride = (kCoopThreads / 64) * kCoopGroups;
    const uint32_t t_first = (c0 << 6) + (threadIdx.x & 63u);
    Step st_next = steps[0];
    LanePos lp_next = classify<1, 0, 0>(st_next, t_first, 0u, 0u);
    GateDesc d_next = lp_next.kind ? descs[st_next.first + lp_next.g] : GateDesc{0, 0, 0, 0};
    uint4 tab_next = eval_tab_row(lp_next, d_next, T);
    coop_gather(xp, W, ninputs, rank, ctl);
    for (uint32_t lv = 0; lv < nsteps; lv++) {
        const Step st = st_next;
        const uint32_t chunks = (level_lanes<1, 0, 0>(st, 0u) + 63u) >> 6;
        if (c0 < chunks)
            eval_group<NR, false, 1, true, true, true>(st, 0u, lp_next.kind, lp_next.g, lp_next.inst, lp_next.q, d_next.in0, d_next.in1,
                                                       d_next.tweak, d_next.row_op, descs, ninputs, 0u, 0u, 1u, W, T, rkr, te, lo,
                                                       pacc, plast, kCoopThreads, tab_next);
        for (uint32_t c = c0 + stride; c < chunks; c += stride)
            eval_group<NR, false, 1, false, true>(st, (c - wave) << 6, 0, 0u, 0u, 0u, 0u, 0u, 0u, 0u, descs, ninputs, 0u, 0u, 1u, W, T,
                                                  rkr, te, lo, pacc, plast, kCoopThreads);
        if (lv + 1 < nsteps) {
            coop_arrive(ctl);
            st_next = steps[lv + 1];
            lp_next = classify<1, 0, 0>(st_next, t_first, 0u, 0u);
            if (lp_next.kind) d_next = descs[st_next.first + lp_next.g];
            // the table row too: it is constant during the pass and comes from HBM (the rows were copied in, not written
            // by this pass), the longest round trip of a level — under the barrier wait instead of behind it
            tab_next = eval_tab_row(lp_next, d_next, T);
            coop_wait(ctl, lv + 1 + (xp ? 1u : 0u));
        }
    }
    coop_scatter(xp, W, rank, ctl, nsteps + 1);
    coop_leave(ctl);
}

// Are the kCoopGroups workgroups on one XCD, and does a value stored before the barrier arrive behind it?  64 rounds of:
// every group writes a fresh value, barrier, reads its neighbour's, barrier.  ctl->bad counts what went wrong.
__global__ __launch_bounds__(kCoopThreads) void k_coop_selftest(CoopCtl *ctl) {
    if (blockIdx.x & 7) return;
    const uint32_t rank = blockIdx.x >> 3;
    if (coop_dropped(ctl, rank)) return;
    uint32_t *scratch = ctl->scratch;  // written with plain stores and read past the L1, as the labels are
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu;  // HW_REG_XCC_ID
    if (threadIdx.x == 0) scratch[32 + rank] = xcc;
    uint32_t bar = 0, bad = 0;
    const uint64_t t_begin = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < 64; it++) {
        if (threadIdx.x == 0) scratch[rank] = it * 131u + rank;
        coop_barrier(ctl, ++bar);
        const uint32_t nb = (rank + 1 + it) % kCoopGroups;
        if (threadIdx.x == 64 && __hip_atomic_load(&scratch[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != it * 131u + nb) bad++;
        coop_barrier(ctl, ++bar);
    }
    if (threadIdx.x == 0 && rank == 0) ctl->ticks = (uint32_t)(__builtin_amdgcn_s_memtime() - t_begin);  // ticks of 128 barriers
    if (threadIdx.x == 64) {
        for (uint32_t r = 0; r < kCoopGroups; r++)
            if (__hip_atomic_load(&scratch[32 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc) bad++;
        if (bad) __hip_atomic_fetch_add(&ctl->bad, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

void launch_coop_selftest(CoopCtl *ctl, hipStream_t s) {
    (void)hipMemsetAsync(ctl, 0, sizeof(CoopCtl), s);
    hipLaunchKernelGGL(k_coop_selftest, dim3(8 * kCoopGroups), dim3(kCoopThreads), 0, s, ctl);
}

void launch_coop(bool eval, const FusedArgs &a, CoopCtl *ctl, const StoreXchg *x, uint32_t seq, hipStream_t s) {
    if (a.nsteps == 0) return;
    // ctl->count is zero here: the self-test's host code and every pass leave it so (coop_leave)
    const dim3 grid(8 * (kCoopGroups + 1)), block(kCoopThreads);  // + the stand-by, on the same XCD
#define GC_CO(NR)                                                                                                          \
    if (eval)                                                                                                              \
        hipLaunchKernelGGL((k_eval_coop<NR>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs, a.W,               \
                           (const uint4 *)a.T, a.rk, a.te0, ctl, x, seq);                                                 \
    else                                                                                                                   \
        hipLaunchKernelGGL((k_garble_coop<NR>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs, a.W, a.R, a.T,   \
                           a.rk, a.te0, ctl, x, seq)
    switch (a.rounds) {
    case 10: GC_CO(10); break;
    case 12: GC_CO(12); break;
    default: GC_CO(14); break;
    }
#undef GC_CO
}

// passes (1024-lane workgroups) of a level for one instance: the measure of "wide" (plan.h: wide_for_one_instance)
uint32_t level1_passes(const Step &st, bool eval) {
    const uint32_t lanes = eval ? (st.n_and << 1) + st.n_or + st.n_inv + (st.count - st.nonfree)
                                : ((st.n_and + st.n_or) << 2) + (st.n_inv << 1) + (st.count - st.nonfree);
    return (lanes + kFusedThreads - 1) / kFusedThreads;
}
// workgroups of a level launch (kLevelLanes lanes each)
static uint32_t level1_groups(const Step &st, bool eval) {
    const uint32_t lanes = eval ? (st.n_and << 1) + st.n_or + st.n_inv + (st.count - st.nonfree)
                                : ((st.n_and + st.n_or) << 2) + (st.n_inv << 1) + (st.count - st.nonfree);
    return (lanes + kLevelLanes - 1) / kLevelLanes;
}

// one launch per level of `levels` (host copy of the device step array a.steps)
void launch_levels1(bool eval, const FusedArgs &a, const Step *levels, hipStream_t s) {
    for (uint32_t lv = 0; lv < a.nsteps; lv++) {
        const Step &st = levels[lv];
        const uint32_t grid = level1_groups(st, eval);
        if (grid == 0) continue;
#define GC_L1(NR)                                                                                                      \
    if (eval)                                                                                                          \
        hipLaunchKernelGGL((k_eval_level1<NR>), dim3(grid), dim3(kFusedThreads), 0, s, a.descs, st, a.ninputs, a.W,    \
                           (const uint4 *)a.T, a.rk, a.te0);                                                           \
    else                                                                                                               \
        hipLaunchKernelGGL((k_garble_level1<NR>), dim3(grid), dim3(kFusedThreads), 0, s, a.descs, st, a.ninputs, a.W,  \
                           a.R, a.T, a.rk, a.te0)
        switch (a.rounds) {
        case 10: GC_L1(10); break;
        case 12: GC_L1(12); break;
        default: GC_L1(14); break;
        }
#undef GC_L1
    }
}

static uint32_t tune_env() {
    const char *e = getenv("GC_TUNE");
    return e ? (uint32_t)atoi(e) : 0u;
}

// every level of the circuit fits one or two column-sliced passes at this tile size (garbler lanes; the evaluator's are fewer)
bool fused_col_form_fits(const Step *levels, uint32_t nsteps, uint32_t ti_log2) {
    uint32_t hashed = 0;
    for (uint32_t i = 0; i < nsteps; i++) {
        const Step &st = levels[i];
        if (st.n_or) return false;
        const uint64_t lanes = ((((uint64_t)st.n_and << 2) + ((uint64_t)st.n_inv << 1)) << 2) + (st.count - st.nonfree);
        if ((lanes << ti_log2) > 2u * (uint64_t)kFusedThreads) return false;  // (the kernels loop over a second pass)
        hashed += st.nonfree;
    }
    return hashed != 0;  // (a circuit without hashed gates gains nothing from the form)
}

void launch_garble_fused(const FusedArgs &a, const BatchGeom &g, hipStream_t s) {
    if (a.nsteps == 0) return;
    dim3 grid(g.ntiles), block(kFusedThreads);
    if (a.narrow_col && !a.prof) {
#define GC_GN(NR)                                                                                                          \
    hipLaunchKernelGGL((k_garble_col<NR>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs, g.ti_log2,            \
                       g.lw.tile_stride, g.lt.tile_stride, a.W, a.R, a.T, a.rk, a.te0)
        switch (a.rounds) {
        case 10: GC_GN(10); break;
        case 12: GC_GN(12); break;
        default: GC_GN(14); break;
        }
#undef GC_GN
        return;
    }
#define GC_GF(NR)                                                                                               \
    if (a.prof)                                                                                                 \
        hipLaunchKernelGGL((k_garble_fused<NR, true>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs,    \
                           g.ti_log2, g.lw.tile_stride, g.lt.tile_stride, a.W, a.R, a.T, a.rk, a.te0, a.prof, tune_env());  \
    else                                                                                                        \
        hipLaunchKernelGGL((k_garble_fused<NR, false>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs,   \
                           g.ti_log2, g.lw.tile_stride, g.lt.tile_stride, a.W, a.R, a.T, a.rk, a.te0, a.prof, tune_env())
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
    if (a.narrow_col && !a.prof) {
#define GC_EN(NR)                                                                                                          \
    hipLaunchKernelGGL((k_eval_col<NR>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs, g.ti_log2,              \
                       g.lw.tile_stride, g.lt.tile_stride, a.W, (const uint4 *)a.T, a.rk, a.te0)
        switch (a.rounds) {
        case 10: GC_EN(10); break;
        case 12: GC_EN(12); break;
        default: GC_EN(14); break;
        }
#undef GC_EN
        return;
    }
#define GC_EF(NR)                                                                                               \
    if (a.prof)                                                                                                 \
        hipLaunchKernelGGL((k_eval_fused<NR, true>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs,      \
                           g.ti_log2, g.lw.tile_stride, g.lt.tile_stride, a.W, (const uint4 *)a.T, a.rk, a.te0, \
                           a.prof, tune_env());                                                                             \
    else                                                                                                        \
        hipLaunchKernelGGL((k_eval_fused<NR, false>), grid, block, 0, s, a.descs, a.steps, a.nsteps, a.ninputs,     \
                           g.ti_log2, g.lw.tile_stride, g.lt.tile_stride, a.W, (const uint4 *)a.T, a.rk, a.te0, \
                           a.prof, tune_env())
    switch (a.rounds) {
    case 10: GC_EF(10); break;
    case 12: GC_EF(12); break;
    default: GC_EF(14); break;
    }
#undef GC_EF
}

}  // namespace gc
