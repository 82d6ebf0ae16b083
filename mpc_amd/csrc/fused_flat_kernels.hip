// fused_flat_kernels.hip — fused, LDS-resident, FLATTENED schedule: the production garble / eval kernels.
//
// One launch per pass; a 1024-thread workgroup owns a tile of TI instances; wire labels live in recycled LDS
// slots next to the perm-addressed AES table.  The plan (plan.h, "flattened schedule") has already expanded
// every XOR output that is needed into a list of chunk-entry labels, so a pass is
//
//     for every unit:   hash part  (all table-producing gates of a hash phase: 2-4 lanes per gate-instance,
//                                   joined by DPP; LDS/VALU-bound AES)
//                       barrier
//                       XOR part   (one lane per needed XOR output and instance: XOR of its term list)
//                       barrier
//
// i.e. two workgroup barriers per hash phase and NO serial XOR chain (the level-by-level XOR walk of
// fused_lds_kernels.hip cost ~20 % of a pass in dependent LDS round trips with the AES pipe idle).
// The next unit's image (descriptors + XOuts, <= 14.5 KiB) is fetched into one register per thread
// while the current unit executes and committed to the other LDS buffer before the closing barrier.
// LDS map: 64 KiB AES table | 2 x ustride uint4 stage (ustride = the circuit's largest unit, <= 14.5 KiB) | R[TI] |
// wires [slot][TI] (last slot = zero label).
#include "aes_device.h"
#include "kernels.h"

namespace gc {

namespace {

constexpr int TF = 1024;

constexpr int DPP_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
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

// v ^ (v of the quad partner): computed by every lane of the quad so that it is four v_xor_b32_dpp; the opaque asm
// keeps the compiler from sinking the XOR into the q == 0 branch that uses it (DPP cannot read from lanes that are
// switched off, so there it becomes 4 DPP moves + 4 XORs)
template <int CTRL>
__device__ __forceinline__ uint4 quad_xor(uint4 v) {
    uint4 o = make_uint4(v.x ^ dpp32<CTRL>(v.x), v.y ^ dpp32<CTRL>(v.y), v.z ^ dpp32<CTRL>(v.z), v.w ^ dpp32<CTRL>(v.w));
    asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(o.w));
    return o;
}

typedef uint32_t lds_v4 __attribute__((ext_vector_type(4)));
using lds_v4p = __attribute__((address_space(3))) const lds_v4 *;
using lds_v4w = __attribute__((address_space(3))) lds_v4 *;

// Wire labels in LDS by byte address: label of slot s, instance inst = ib + (s << sh) with ib = wl + 16 inst and
// sh = ti_log2 + 4: one v_lshl_add_u32 per access (the indexed form costs two shifts and a three-input add)
__device__ __forceinline__ uint4 lds_label(uint32_t ib, uint32_t slot, uint32_t sh) {
    const lds_v4 v = *(lds_v4p)(uintptr_t)((slot << sh) + ib);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_label_put(uint32_t ib, uint32_t slot, uint32_t sh, uint4 v) {
    lds_v4 o;
    o.x = v.x, o.y = v.y, o.z = v.z, o.w = v.w;
    *(lds_v4w)(uintptr_t)((slot << sh) + ib) = o;
}
// low (hi = false) or high half-word of a packed slot pair, chosen per lane: one v_perm_b32
__device__ __forceinline__ uint32_t half_of(uint32_t packed, bool hi) {
    return __builtin_amdgcn_perm(packed, packed, hi ? 0x0c0c0302u : 0x0c0c0100u);
}

// h ^ (w & m) per bit: one v_bitop3_b32 per word
__device__ __forceinline__ uint4 xand4(uint4 h, uint4 w, uint32_t m) {
    return make_uint4(__builtin_amdgcn_bitop3_b32(h.x, w.x, m, 0x78), __builtin_amdgcn_bitop3_b32(h.y, w.y, m, 0x78),
                      __builtin_amdgcn_bitop3_b32(h.z, w.z, m, 0x78), __builtin_amdgcn_bitop3_b32(h.w, w.w, m, 0x78));
}

__device__ __forceinline__ uint4 lxor3(uint4 a, uint4 b, uint4 c) {
    return make_uint4(xor3(a.x, b.x, c.x), xor3(a.y, b.y, c.y), xor3(a.z, b.z, c.z), xor3(a.w, b.w, c.w));
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wide-form hash lanes up to which a unit takes the column-sliced form (4 x as many lanes: one pass of the workgroup)
constexpr uint32_t kNarrowLanes = 256;
// largest column-sliced tail behind full sets of four wide waves (blocks): beyond it a fourth quarter-cost wave per
// SIMD costs more than the one wide wave it replaces (measured: 128 and 192 are equal, 64 loses 1 %)
constexpr uint32_t kTailMax = 192;
// the garbler's column lanes do more around the AES (R, both permute bits, two table rows): its optimum is lower
// (measured 0.637 / 0.627 / 0.627 / 0.630 ms for 0 / 64 / 128 / 192)
constexpr uint32_t kTailMaxGarble = 128;
constexpr uint32_t kKeyTab = kTeDualBytes;            // byte address of the round-key table of the column-sliced hashes
constexpr uint32_t kStageOff = kTeDualBytes / 16 + 16;  // 256 bytes: 15 round keys x 4 columns
static_assert(kStageOff == kFlatStageOff16, "plan.h: the planner's copy of the LDS geometry");

#define GC_FPROF(slot)                                               \
    if constexpr (PROF) {                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
        const uint64_t now__ = __builtin_amdgcn_s_memtime();         \
        pacc[slot] += now__ - plast;                                 \
        plast = now__;                                               \
    }
// debug cycle breakdown (gc_batch_debug_profile): 0 unit header / prefetch issue, 1 hash part, 2 barrier A,
// 3 commit, 4 XOR part, 5 barrier B; written for wave 0 (counters 0-7) and wave 15 (8-15) of every workgroup
#define GC_FPROF_EPILOGUE()                                                                                  \
    if constexpr (PROF) {                                                                                    \
        if (threadIdx.x == 0 || threadIdx.x == 960)                                                          \
            for (int i = 0; i < 8; i++) a.prof[(size_t)tile * 16 + (threadIdx.x ? 8 : 0) + i] = pacc[i];       \
    }

using FlArgs = FlatJob;  // kernels.h

// Unit header i.  Loaded with VECTOR loads on purpose (vz is a zero the compiler cannot see through): scalar loads
// share the lgkm counter with LDS and return out of order, so with a header in flight the first LDS read of the
// next unit would have to wait for it (s_waitcnt lgkmcnt(0)); vector loads are tracked by vmcnt and cost nothing
// until their values are used one unit later.
// The unit array ends with two all-zero records and the program with one stage buffer of padding (engine.cpp), so
// headers and images two units past the end can be fetched without bounds checks (they describe empty units).
__device__ __forceinline__ FUnit load_unit(const FUnit *units, uint32_t i, uint32_t vz) {
    FUnit u;
    const uint4 *p = (const uint4 *)(units + i) + vz;
    const uint4 a = p[0], b = p[1], c = p[2];
    u.off16 = a.x, u.n16 = a.y, u.n_and = a.z, u.n_or = a.w;
    u.n_inv = b.x, u.nout = b.y, u.outs_off16 = b.z, u.xparts = b.w;
    u.hfirst = c.x, u.ofirst = c.y;
    u.pad_[0] = u.pad_[1] = 0;
    return u;
}
// wave-uniform copy in SGPRs (readfirstlane) of a header whose loads have landed
__device__ __forceinline__ FUnit uniform_unit(const FUnit &v) {
    FUnit u{};
    u.off16 = __builtin_amdgcn_readfirstlane(v.off16);
    u.n16 = __builtin_amdgcn_readfirstlane(v.n16);
    u.n_and = __builtin_amdgcn_readfirstlane(v.n_and);
    u.n_or = __builtin_amdgcn_readfirstlane(v.n_or);
    u.n_inv = __builtin_amdgcn_readfirstlane(v.n_inv);
    u.nout = __builtin_amdgcn_readfirstlane(v.nout);
    u.outs_off16 = __builtin_amdgcn_readfirstlane(v.outs_off16);
    u.hfirst = __builtin_amdgcn_readfirstlane(v.hfirst);
    u.ofirst = __builtin_amdgcn_readfirstlane(v.ofirst);
    u.xparts = __builtin_amdgcn_readfirstlane(v.xparts);
    return u;
}

// hash-part lane -> (kind, gate, instance, sub-lane); kinds: 0 none, 1 AND, 2 OR, 3 INV
struct HP {
    uint32_t kind, g, inst, q;
};
template <int LQA, int LQO, int LQI, bool HAS_OR = true>
__device__ __forceinline__ HP hpos(uint32_t t, const FUnit &c, uint32_t ti_log2, uint32_t tim) {
    HP p{0, 0, 0, 0};
    if constexpr (!HAS_OR) {  // AND lanes, then INV lanes: branch-free (selects instead of exec-mask regions)
        const uint32_t e_and = (c.n_and << ti_log2) << LQA;
        const uint32_t e_all = e_and + ((c.n_inv << ti_log2) << LQI);
        const bool is_and = t < e_and;
        const uint32_t u = is_and ? t : t - e_and;
        const uint32_t lq = is_and ? (uint32_t)LQA : (uint32_t)LQI;
        p.kind = is_and ? 1u : t < e_all ? 3u : 0u;
        p.g = (u >> (ti_log2 + lq)) + (is_and ? 0u : c.n_and);
        p.inst = (u >> lq) & tim;
        p.q = u & ((1u << lq) - 1);
        return p;
    }
    const uint32_t e_and = (c.n_and << ti_log2) << LQA;
    const uint32_t e_or = HAS_OR ? e_and + ((c.n_or << ti_log2) << LQO) : e_and;
    const uint32_t e_all = e_or + ((c.n_inv << ti_log2) << LQI);
    if (t < e_and) {
        p.kind = 1;
        p.g = t >> (ti_log2 + LQA);
        p.inst = (t >> LQA) & tim;
        p.q = t & ((1u << LQA) - 1);
    } else if (HAS_OR && t < e_or) {
        const uint32_t u = t - e_and;
        p.kind = 2;
        p.g = c.n_and + (u >> (ti_log2 + LQO));
        p.inst = (u >> LQO) & tim;
        p.q = u & ((1u << LQO) - 1);
    } else if (t < e_all) {
        const uint32_t u = t - e_or;
        p.kind = 3;
        p.g = c.n_and + c.n_or + (u >> (ti_log2 + LQI));
        p.inst = (u >> LQI) & tim;
        p.q = u & ((1u << LQI) - 1);
    }
    return p;
}
template <int LQA, int LQO, int LQI>
__device__ __forceinline__ uint32_t hlanes(const FUnit &c, uint32_t ti_log2) {
    return ((c.n_and << ti_log2) << LQA) + ((c.n_or << ti_log2) << LQO) + ((c.n_inv << ti_log2) << LQI);
}

// value of the lane SH further on inside its row of 16 lanes (DPP row_shl: register to register, no LDS)
template <int SH>
__device__ __forceinline__ uint4 row_down(uint4 v) {
    return make_uint4((uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x100 + SH, 0xf, 0xf, true),
                      (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x100 + SH, 0xf, 0xf, true),
                      (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x100 + SH, 0xf, 0xf, true),
                      (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x100 + SH, 0xf, 0xf, true));
}
__device__ __forceinline__ uint4 sel4(bool c, uint4 a, uint4 b) {
    return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
// leader of a list spread over 2 / 4 lanes TI apart (all inside one row: parts * TI <= 16, plan.h / geom_for): its own
// partial sum plus those of the following parts; every other lane keeps its value
template <int TIV>
__device__ __forceinline__ uint4 join_parts(uint4 acc, uint32_t flags, bool four) {
    const uint4 a1 = lxor(acc, row_down<TIV>(acc));
    uint4 r = sel4((flags & (kXoJoin2 | kXoJoin4)) != 0, a1, acc);
    if constexpr (2 * TIV < 16) {
        if (four) {
            const uint4 a2 = lxor(a1, row_down<2 * TIV>(a1));
            r = sel4((flags & kXoJoin4) != 0, a2, r);
        }
    }
    return r;
}

// XOR part: one lane per (XOut, instance): the label is the XOR of up to 8 terms whose LDS slots sit in the XOut
// itself (one LDS round trip for the item, one for its labels); lists of more than 8 terms come as 2 / 4 parts.
// garble.go:331-351 / eval.go:49-51 restated over the expanded terms.
template <bool GARBLE>
__device__ __forceinline__ void xor_part(const uint4 *buf, const FUnit &u, const uint32_t *ogslot, uint4 *wl,
                                         const uint4 *rl, uint4 *Wt, uint32_t ti_log2, uint32_t tim) {
    const uint2 *outs = (const uint2 *)(buf + u.outs_off16);  // 3 x 8 bytes per XOut
    const uint32_t nitems = u.nout << ti_log2;
    for (uint32_t t = threadIdx.x; t < nitems; t += TF) {
        const uint32_t o = t >> ti_log2, inst = t & tim;
        const uint2 d0 = outs[3 * o], d2 = outs[3 * o + 2];
        uint2 d1 = outs[3 * o + 1];
        // keep the read of the second half of the item with the other two (one LDS round trip for the whole item);
        // the compiler would otherwise sink it into the n > 4 branch: a third dependent round trip for long lists
        asm volatile("" : "+v"(d1.x), "+v"(d1.y));
        const uint32_t flags = d2.x >> 16, n = d2.y & 0xffffu;
        // byte addressing by hand: label of slot s, instance inst = wl + (s << (ti_log2 + 4)) + 16 inst — one shift of
        // the extracted half-word and one add per term
        const uint32_t sh = ti_log2 + 4, ib = (uint32_t)(uintptr_t)wl + (inst << 4);
        auto lab = [&](uint32_t packed, bool high) {
            const uint32_t s16 = high ? packed >> 16 : packed & 0xffffu;
            const lds_v4 v = *(lds_v4p)(uintptr_t)((s16 << sh) + ib);
            return make_uint4(v.x, v.y, v.z, v.w);
        };
        // Items are sorted by length and padded with the zero slot: a wave that holds a list of more than four terms
        // reads all eight labels in ONE batch (a wave-uniform branch; the short lists of that wave read zeros), the
        // other waves read four.  Three-input XORs (v_bitop3): 2 / 4 per word.
        uint4 acc;
        if (__ballot(n > 4) != 0) {
            const uint4 v0 = lab(d0.x, false), v1 = lab(d0.x, true), v2 = lab(d0.y, false), v3 = lab(d0.y, true);
            const uint4 v4 = lab(d1.x, false), v5 = lab(d1.x, true), v6 = lab(d1.y, false), v7 = lab(d1.y, true);
            acc = lxor(lxor3(lxor3(v0, v1, v2), v3, v4), lxor3(v5, v6, v7));
        } else {
            const uint4 v0 = lab(d0.x, false), v1 = lab(d0.x, true), v2 = lab(d0.y, false), v3 = lab(d0.y, true);
            acc = lxor(lxor3(v0, v1, v2), v3);
        }
        // collect the partial sums of lists that were spread over 2 / 4 lanes: only in units that have such lists, and
        // there only in the waves that hold them (they come first in the length order) - a wave-uniform test
        if (u.xparts > 1 && __ballot((flags & (kXoJoin2 | kXoJoin4 | kXoPart)) != 0) != 0) {
            const bool four = u.xparts > 2;
            if (ti_log2 == 0) acc = join_parts<1>(acc, flags, four);
            else if (ti_log2 == 1) acc = join_parts<2>(acc, flags, four);
            else if (ti_log2 == 2) acc = join_parts<4>(acc, flags, four);
            else acc = join_parts<8>(acc, flags, false);
        }
        if (flags & kXoPart) continue;
        if (GARBLE && (flags & kXoRpar)) acc = lxor(acc, rl[inst]);
        lds_label_put(ib, d2.x & 0xffffu, sh, acc);
        if (flags & kXoStore) Wt[((size_t)ogslot[u.ofirst + o] << ti_log2) + inst] = acc;
    }
}

// How the hash lanes of a unit are split between the two forms.  The wide form costs one wave per 64 lanes and a hash
// phase lasts as long as the busiest SIMD: with L lanes in the last pass of the workgroup, L = 256 k + r, the r lanes
// past the last full set of four waves cost a whole extra wave on one SIMD.  Those r blocks go column-sliced instead
// (4 r lanes on the waves the wide form leaves idle in that pass, ceil(r / 64) quarter-cost waves on EVERY SIMD) when
// that is cheaper: always if the pass has no full set (k = 0: the former "narrow unit"), else for r <= 192 if the idle
// waves suffice.  Units with an OR gate stay wide.
struct HashSplit {
    uint32_t wide_end;   // hash lanes [0, wide_end) run wide (a multiple of 256 when a tail exists)
    uint32_t tail;       // number of blocks (wide lanes) in the column-sliced tail, 0: none
    uint32_t tail_tid;   // first thread of the tail's column lanes
};
// branch-free on purpose (selects on SGPRs): it runs in every wave at the head of every unit, and as early-return code
// it compiled to ~15 scalar branches — 200 cycles per unit
__device__ __forceinline__ HashSplit split_hash_lanes(uint32_t e_all, bool has_or_gate, uint32_t tail_max = kTailMax) {
    const uint32_t rem = e_all & (TF - 1u);  // lanes of the last, partial pass (0: the passes are all full)
    const bool small = rem <= kNarrowLanes;
    const uint32_t k = small ? 0u : rem >> 8, r = small ? rem : rem & 255u;
    const bool fits = (k == 0) | ((r <= tail_max) & (((r + 15u) >> 4) + 4u * k <= 16u));
    const uint32_t tail = (!has_or_gate & fits) ? r : 0u;
    HashSplit h;
    h.tail = tail;
    h.wide_end = e_all - tail;
    h.tail_tid = h.wide_end & (TF - 1u);
    return h;
}

// ---- column-sliced hash lanes (narrow units and the tails of wide ones) ----------------------------------------
// Every hash lane of the wide form becomes a quad (lane = 4 * wide lane + column): a lone wave's 14 AES rounds take
// ~3.3 k cycles, the four quarter-waves of the column form ~2.2 k, and in such units that latency is the phase.  Same
// arithmetic as the wide form, one 32-bit column per lane: label word W_c (big-endian column c) sits at dword c ^ 1.
// Partner lanes of a gate are 4 (q ^ 1) and 8 (q ^ 2) lanes away: DPP row shifts / rotations inside the row of 16.
__device__ __forceinline__ uint32_t lds_word(uint32_t addr) { return *(lds_u32 *)(uintptr_t)addr; }
__device__ __forceinline__ void lds_word_put(uint32_t addr, uint32_t v) {
    *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)addr = v;
}
// value of the lane 4 further on (q even) / 4 back (q odd): the q ^ 1 partner
__device__ __forceinline__ uint32_t pair4(uint32_t v) {
    uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0x5, false);   // row_shl:4 -> banks 0, 2
    return (uint32_t)__builtin_amdgcn_update_dpp((int)r, (int)v, 0x114, 0xf, 0xa, false);     // row_shr:4 -> banks 1, 3
}
// column c of K ^ rk_0 with K = 2x ^ tweak: (x_c << 1) | (x_c+1 >> 31), the tweak in column 3
__device__ __forceinline__ uint32_t whiten_col(uint32_t xc, uint32_t xc1, uint32_t c, uint32_t tweak, uint32_t k0) {
    const uint32_t kcol = __builtin_amdgcn_alignbit(xc, c == 3 ? 0u : xc1, 31);
    return xor3(kcol, c == 3 ? tweak : 0u, k0);
}

template <int NR, bool PROF = false>
__device__ __forceinline__ void garble_hash_narrow(const uint4 *buf, const FUnit &u, const FlArgs &a, uint32_t ti_log2,
                                                   uint32_t tim, uint4 *wl, const uint4 *rl, uint4 *Tt, uint4 *Wt,
                                                   uint32_t lo, const HashSplit hs, uint64_t *pacc = nullptr,
                                                   uint64_t *plast_p = nullptr) {
    uint64_t plast = PROF ? *plast_p : 0;
    const uint32_t TI = 1u << ti_log2;
    const uint32_t j = threadIdx.x - hs.tail_tid;  // column lane j = 4 * (block of the tail) + column
    if (threadIdx.x < hs.tail_tid || j >= 4u * hs.tail) return;
    const HP hp = hpos<2, 2, 1, false>(hs.wide_end + (j >> 2), u, ti_log2, tim);
    const uint32_t c = j & 3u, q = hp.q, inst = hp.inst, wo = (c ^ 1u) << 2, wo1 = ((c + 1u) ^ 1u) << 2;
    const uint4 dv = buf[hp.g];
    const FDesc d{dv.x, dv.y, dv.z, dv.w};
    const uint32_t sh = ti_log2 + 4, ib = (uint32_t)(uintptr_t)wl + (inst << 4);
    const uint32_t ra = (uint32_t)(uintptr_t)rl + (inst << 4);
    // the two operands' labels (an INV has one: its second slot field is not a slot)
    const uint32_t sa = ((d.lin & 0xffffu) << sh) + ib, sb = hp.kind == 1 ? ((d.lin >> 16) << sh) + ib : sa;
    const uint32_t so = (q & 2u) ? sb : sa;                                                // INV lanes have q < 2
    const uint32_t keyaddr = kKeyTab + (c << 2);
    // one LDS round trip: own operand (columns c, c+1), R (c, c+1), a0 column c, the two permute bits, round key 0
    const uint32_t bc = lds_word(so + wo), bc1 = lds_word(so + (wo1 & 12u));
    const uint32_t rc = lds_word(ra + wo), rc1 = lds_word(ra + (wo1 & 12u));
    const uint32_t a0c = lds_word(sa + wo), a0y = lds_word(sa + 4), b0y = lds_word(sb + 4);
    const uint32_t k0 = lds_word(keyaddr);
    const uint32_t modd = (q & 1u) ? ~0u : 0u;
    const uint32_t xc = __builtin_amdgcn_bitop3_b32(bc, rc, modd, 0x78), xc1 = __builtin_amdgcn_bitop3_b32(bc1, rc1, modd, 0x78);
    GC_FPROF(6)
    const uint32_t h = hash_col_whitened<NR>(whiten_col(xc, xc1, c, d.tweak + (q >> 1), k0), keyaddr, lo);
    GC_FPROF(7)
    if constexpr (PROF) *plast_p = plast;
    const uint32_t rowb = ((((d.row_op & kRowMask) << ti_log2) + inst) << 4) + wo;
    auto row = [&](uint32_t r) -> uint32_t & { return *(uint32_t *)((char *)Tt + (rowb + (r << 4))); };
    auto put = [&](uint32_t v) {
        if (q == 0) {
            lds_word_put(((d.lout & 0xffffu) << sh) + ib + wo, v);
            if (d.lout & kFStoreGlobal)
                *(uint32_t *)((char *)(Wt + ((size_t)a.hgslot[u.hfirst + hp.g] << ti_log2) + inst) + wo) = v;
        }
    };
    const uint32_t p = h ^ pair4(h);
    if (hp.kind == 1) {  // garble.go:353-395, one column
        const uint32_t m2 = (q & 2u) ? ~0u : 0u;
        const uint32_t pa = (uint32_t)((int32_t)a0y >> 31), pb = (uint32_t)((int32_t)b0y >> 31);
        const uint32_t mk = m2 ? pb : pa, rm = pb & ~m2;
        const uint32_t w = __builtin_amdgcn_bitop3_b32(p, rc, rm, 0x78);
        const uint32_t tab = __builtin_amdgcn_bitop3_b32(w, a0c, m2, 0x78);
        const uint32_t v = __builtin_amdgcn_bitop3_b32(h, w, mk, 0x78);
        if (!(q & 1u)) row((q & 2u) ? TI : 0) = tab;
        // the q ^ 2 partner is 8 lanes away: a rotation of the row by 8
        uint32_t o = v ^ (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, true);
        asm volatile("" : "+v"(o));
        put(o);
    } else {  // INV, garble.go:446-474
        if (q == 0) row(0) = p ^ rc;
        put(h ^ (((int32_t)a0y < 0) ? p : rc));
    }
}

template <int NR, bool PROF = false>
__device__ __forceinline__ void eval_hash_narrow(const uint4 *buf, const FUnit &u, const FlArgs &a, uint32_t ti_log2,
                                                 uint32_t tim, uint4 *wl, const uint4 *Tt, uint4 *Wt, uint32_t lo,
                                                 const HashSplit hs, uint64_t *pacc = nullptr, uint64_t *plast_p = nullptr) {
    uint64_t plast = PROF ? *plast_p : 0;
    const uint32_t TI = 1u << ti_log2;
    const uint32_t j = threadIdx.x - hs.tail_tid;
    if (threadIdx.x < hs.tail_tid || j >= 4u * hs.tail) return;
    const HP hp = hpos<1, 0, 0, false>(hs.wide_end + (j >> 2), u, ti_log2, tim);
    const uint32_t c = j & 3u, q = hp.q, inst = hp.inst, wo = (c ^ 1u) << 2, wo1 = ((c + 1u) ^ 1u) << 2;
    const uint4 dv = buf[hp.g];
    const FDesc d{dv.x, dv.y, dv.z, dv.w};
    const uint32_t sh = ti_log2 + 4, ib = (uint32_t)(uintptr_t)wl + (inst << 4);
    const uint32_t sa = ((d.lin & 0xffffu) << sh) + ib, sb = hp.kind == 1 ? ((d.lin >> 16) << sh) + ib : sa;
    const uint32_t so = q ? sb : sa;  // AND lane 1 hashes operand b (INV: q = 0)
    const uint32_t keyaddr = kKeyTab + (c << 2);
    const uint32_t rowb = ((((d.row_op & kRowMask) << ti_log2) + inst) << 4) + wo;
    const uint32_t tab = *(const uint32_t *)((const char *)Tt + (rowb + ((q ? TI : 0u) << 4)));  // lands during the AES
    const uint32_t xc = lds_word(so + wo), xc1 = lds_word(so + (wo1 & 12u)), xy = lds_word(so + 4);
    const uint32_t ac = lds_word(sa + wo), k0 = lds_word(keyaddr);
    GC_FPROF(6)
    const uint32_t h = hash_col_whitened<NR>(whiten_col(xc, xc1, c, d.tweak + q, k0), keyaddr, lo);
    GC_FPROF(7)
    if constexpr (PROF) *plast_p = plast;
    const uint32_t sm = (uint32_t)((int32_t)xy >> 31);
    auto put = [&](uint32_t v) {
        if (q == 0) {
            lds_word_put(((d.lout & 0xffffu) << sh) + ib + wo, v);
            if (d.lout & kFStoreGlobal)
                *(uint32_t *)((char *)(Wt + ((size_t)a.hgslot[u.hfirst + hp.g] << ti_log2) + inst) + wo) = v;
        }
    };
    if (hp.kind == 1) {  // eval.go:53-78: lane 0 WG = H(a) ^ (sa ? TG : 0), lane 1 WE = H(b) ^ (sb ? TE ^ a : 0)
        const uint32_t v = __builtin_amdgcn_bitop3_b32(h, __builtin_amdgcn_bitop3_b32(tab, ac, q ? ~0u : 0u, 0x78), sm, 0x78);
        uint32_t o = v ^ (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x104, 0xf, 0xf, true);  // + the lane 4 further on (q = 1)
        asm volatile("" : "+v"(o));
        put(o);
    } else {  // eval.go:96-109
        put(__builtin_amdgcn_bitop3_b32(h, tab, sm, 0x78));
    }
}

// The stream's wire store, as the job launches read and write it: a label may have been written by ANOTHER workgroup of the
// same launch — a unit this one waited for (unit_enter below), on another XCD, behind another L2 — so its loads and stores
// are agent-scope accesses (sc1: past the L1, and coherent across the XCDs' L2s), and neither side needs to write back or
// invalidate a whole L2 (buffer_wbl2 / buffer_inv sc1: ~12 us each, measured for the cooperative passes).
__device__ __forceinline__ uint4 store_get(const uint4 *p) {
    const uint64_t *q = (const uint64_t *)p;
    const uint64_t lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__device__ __forceinline__ void store_put(uint4 *p, const uint4 &v) {
    uint64_t *q = (uint64_t *)p;
    __hip_atomic_store(q, (uint64_t)v.x | ((uint64_t)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, (uint64_t)v.z | ((uint64_t)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr uint32_t kDepSpins = 1u << 22;  // bound of a wait for another unit (~2 s), in polls
// a word of an upload region that came up by DMA while this kernel was already running (workgroups that serve units of later
// launches: kernels.h: PoolCtl): past this XCD's L2, which is coherent with memory only at kernel boundaries
__device__ __forceinline__ uint32_t fresh_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Dataflow across launches (kernels.h: DfBlock): wait until the word at p has reached `want` (wrap-safe), bounded by polls
__device__ __forceinline__ void df_wait_reached(const uint32_t *p, uint32_t want, uint32_t *host_err) {
    for (uint32_t spins = 0; (int32_t)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0;) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > kDepSpins) {
            if (host_err) __hip_atomic_fetch_max(host_err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}

// MULTI (job launches): `a` is this workgroup's own record, its tile is 0 and the input labels come from the wire store
#define GC_FL_PROLOGUE(LOAD_R)                                                                               \
    const uint32_t tile = MULTI ? 0u : blockIdx.x;                                                           \
    extern __shared__ uint4 smem[];                                                                          \
    uint32_t *te = (uint32_t *)smem;                                                                         \
    const uint32_t ti_log2 = a.ti_log2, TI = 1u << ti_log2, tim = TI - 1;                                    \
    uint4 *stage = smem + kStageOff;                                                                         \
    const uint32_t ustride = a.ustride;                                                                      \
    uint4 *rl = smem + kStageOff + 2 * ustride;                                                              \
    uint4 *wl = rl + TI;                                                                                     \
    /* (a workgroup that runs a CHAIN of jobs, one after the other, keeps the AES table and the column keys of the first) */ \
    const bool first_job = !MULTI || a.pad2_ == 0;                                                           \
    if (first_job) load_te_dual(te, a.te0);                                                                  \
    uint32_t rkr[4 * (NR + 1)];                                                                              \
    uint32_t vz;                                                                                             \
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));                                                              \
    load_round_keys_split<NR, (4 * (NR + 1) > 32 ? 4 * (NR + 1) - 32 : 4 * (NR + 1))>(rkr, a.rk, vz);        \
    fold_last_round_key<NR>(rkr); /* the hashes run on whitened blocks (hash_dual_whitened) */              \
    if (first_job && threadIdx.x < 4 * (NR + 1)) { /* the same keys, column-addressable, for hash_col_whitened */ \
        uint32_t kv = a.rk[threadIdx.x];                                                                     \
        if (threadIdx.x >= 4 * NR) kv ^= a.rk[threadIdx.x - 4 * NR];                                         \
        ((uint32_t *)smem)[kKeyTab / 4 + threadIdx.x] = kv;                                                  \
    }                                                                                                        \
    uint4 *Wt = a.W + (size_t)tile * a.w_tile;                                                               \
    if (threadIdx.x < TI) wl[(a.zslot << ti_log2) + threadIdx.x] = make_uint4(0, 0, 0, 0);                   \
    if (LOAD_R && a.rnd) {                                                                                   \
        /* garbler: draw R and the input zero-labels of the tile straight from the caller's random stream   */ \
        /* ([instance][1 + ninputs] big-endian labels; garble.go:253-258, 271-278) - no separate init kernel */ \
        for (uint32_t i = threadIdx.x; i < ((a.ninputs + 1) << ti_log2); i += TF) {                          \
            const uint32_t j = i >> ti_log2, inst = i & tim, gi = tile * TI + inst;                          \
            uint4 v = make_uint4(0, 0, 0, 0);                                                                \
            if (gi < a.batch) {                                                                              \
                const uint4 raw = a.rnd[(size_t)gi * (a.ninputs + 1) + j];                                   \
                v = make_uint4(__builtin_bswap32(raw.y), __builtin_bswap32(raw.x), __builtin_bswap32(raw.w), \
                               __builtin_bswap32(raw.z));                                                    \
            }                                                                                                \
            if (j == 0) {                                                                                    \
                v.y |= 0x80000000u; /* R.SetS(true) */                                                       \
                rl[inst] = v;                                                                                \
                a.Rout[(size_t)tile * TI + inst] = v;                                                        \
            } else {                                                                                         \
                const uint32_t ls = a.in_lds[j - 1];                                                         \
                Wt[((j - 1) << ti_log2) + inst] = v;                                                         \
                if (ls != 0xffffu) wl[(ls << ti_log2) + inst] = v;                                           \
            }                                                                                                \
        }                                                                                                    \
    } else {                                                                                                 \
        if (LOAD_R && threadIdx.x < TI) rl[threadIdx.x] = a.R[(size_t)tile * TI + threadIdx.x];              \
        for (uint32_t i = threadIdx.x; i < (a.ninputs << ti_log2); i += TF) {                                \
            const uint32_t w = i >> ti_log2, ls = a.in_lds[w];                                               \
            if constexpr (MULTI) { /* Get through in[] (stream_garble.go:131-141) on the device */           \
                const DfBlock *dfb = LOAD_R ? (const DfBlock *)a.prof : nullptr; /* garbler, dataflow across launches: the wire's version first */ \
                const uint32_t sidx = dfb ? fresh_u32(a.in_idx + w) : a.in_idx[w];                           \
                if (dfb) df_wait_reached(dfb->ver + sidx, fresh_u32((const uint32_t *)(dfb + 1) + w), dfb->host_err); \
                const uint4 v = store_get(a.store + sidx);                                                   \
                Wt[i] = v;                                                                                   \
                if (ls != 0xffffu) wl[(ls << ti_log2) + (i & tim)] = v;                                      \
                if (dfb) { /* ... and the read is counted once the label is here (whoever overwrites the wire waits for it) */ \
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         \
                    __hip_atomic_fetch_add(dfb->rd + sidx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  \
                }                                                                                            \
            } else if (ls != 0xffffu) {                                                                      \
                wl[(ls << ti_log2) + (i & tim)] = Wt[i];                                                     \
            }                                                                                                \
        }                                                                                                    \
    }                                                                                                        \
    FUnit u = uniform_unit(load_unit(a.units, 0, vz));                                             \
    /* the next header stays in VGPRs (the kernels are short of SGPRs: 60 hold the AES-256 round keys) and is  */ \
    /* only made wave-uniform when it becomes the current one; headers run two units ahead                   */ \
    FUnit un = load_unit(a.units, 1, vz);                                                          \
    if (threadIdx.x < u.n16) stage[threadIdx.x] = a.prog[u.off16 + threadIdx.x];                             \
    /* unit images run TWO units ahead: the image committed to LDS at the end of unit ui was fetched during unit */ \
    /* ui - 1 (a whole unit of slack: on narrow circuits a unit is shorter than an HBM round trip)            */ \
    uint4 pre_cur = a.prog[un.off16 + threadIdx.x];                                                          \
    __syncthreads();                                                                                         \
    const uint32_t lo = te_lane_off();                                                                       \
    const uint32_t wave_base = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);                           \
    uint64_t pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = 0;                                                  \
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();

}  // namespace

// HAS_OR = false: the circuit has no OR gate (the common case): the OR paths and their selects are compiled out
template <int NR, bool PROF, bool HAS_OR, bool MULTI>
__device__ __forceinline__ void garble_flat_body(const FlArgs &a) {
    GC_FL_PROLOGUE(true)
    uint4 *Tt = a.T + (size_t)tile * a.t_tile;
    for (uint32_t ui = 0; ui < a.nunits; ui++) {
        const uint4 *buf = stage + (ui & 1u) * ustride;
        const uint32_t nh = u.n_and + u.n_or + u.n_inv;
        GC_FPROF(0)
        const uint32_t e_hash = hlanes<2, 2, 1>(u, ti_log2);
        if (nh && u.n_or == 0 && e_hash <= kNarrowLanes) {  // a narrow unit: column-sliced as a whole
            garble_hash_narrow<NR, PROF>(buf, u, a, ti_log2, tim, wl, rl, Tt, Wt, lo, HashSplit{0, e_hash, 0}, pacc, &plast);
        } else if (nh) {
            // wide form for whole sets of four waves, column-sliced form for what is left of the last pass
            const HashSplit hs = split_hash_lanes(e_hash, u.n_or != 0, kTailMaxGarble);
            const uint32_t e_all = hs.wide_end;
        for (uint32_t t0 = 0; t0 + wave_base < e_all; t0 += TF) {  // scalar test: a wave without lanes leaves at once
            const HP hp = hpos<2, 2, 1, HAS_OR>(t0 + threadIdx.x, u, ti_log2, tim);
            if (hp.kind == 0) continue;
            const uint4 dv = buf[hp.g];
            const FDesc d{dv.x, dv.y, dv.z, dv.w};
            const uint32_t inst = hp.inst, q = hp.q;
            const uint4 R = rl[inst];
            // AND q=0..3 hash a0,a1,b0,b1 (lanes 2,3 read operand b and use tweak + 1); INV q=0,1 hash a0,a1
            // (INV lanes have q < 2, so only the OR kind needs the test)
            const bool second = (!HAS_OR || hp.kind == 1) && (q & 2);
            const uint32_t sh = ti_log2 + 4, ib = (uint32_t)(uintptr_t)wl + (inst << 4);
            const uint4 va = lds_label(ib, half_of(d.lin, second), sh);
            uint4 base;
            uint32_t k[4];
            if (HAS_OR && hp.kind == 2) {  // OR: e[2u+v] = enc(a_u, b_v, 0, id)  (garble.go:74-83, 421-424)
                const uint4 vb = lds_label(ib, d.lin >> 16, sh);
                const uint4 x = lxor(va, land(R, (q & 2) ? ~0u : 0u));
                const uint4 y = lxor(vb, land(R, (q & 1) ? ~0u : 0u));
                base = make_uint4(x.y, y.y, 0, 0);
                whiten_k(x, y, d.tweak, rkr, k);
            } else {  // K = 2x ^ tweak, x = the operand's zero label (q even) or one label (q odd)
                base = va;
                const uint4 x = xand4(base, R, (q & 1) ? ~0u : 0u);  // base ^ (q odd ? R : 0): one v_bitop3 per word
                whiten_half(x, d.tweak + (HAS_OR ? (second ? 1u : 0u) : (q >> 1)), rkr, k);
            }
            GC_FPROF(6)  // debug profile: slot 6 = operand fetch + key set-up, slot 7 = AES, slot 1 = combine + stores
            const uint4 h = hash_dual_whitened<NR>(k, rkr, te, lo);
            GC_FPROF(7)
            // 32-bit byte offset into the tile's table rows (launch_fused_flat checks that a tile's rows stay below
            // 4 GiB): scalar base + one VGPR offset instead of 64-bit address arithmetic per lane
            const uint32_t rowb = (((d.row_op & kRowMask) << ti_log2) + inst) << 4;
            auto row = [&](uint32_t r) -> uint4 & { return *(uint4 *)((char *)Tt + (rowb + (r << 4))); };
            // the zero label of the output wire goes out from lane 0 of the gate (called inside every branch: merging
            // the branches' results first costs a register copy per word)
            auto put = [&](const uint4 &out_label) {
                if (q == 0) {
                    lds_label_put(ib, d.lout & 0xffffu, sh, out_label);
                    if (d.lout & kFStoreGlobal)
                        Wt[((size_t)a.hgslot[u.hfirst + hp.g] << ti_log2) + inst] = out_label;
                }
            };
            if (hp.kind == 1) {  // garble.go:353-395; branch-free over the four lanes of the gate (q): bit selects
                const uint32_t m2 = (q & 2) ? ~0u : 0u;        // lanes 2,3 build TE / WE0, lanes 0,1 TG / WG0
                const uint4 p = lxor(h, dpp128<DPP_XOR1>(h));  // lanes 0,1: Ha0^Ha1 ; lanes 2,3: Hb0^Hb1
                const uint4 a0 = dpp128<DPP_BC0>(base);
                const uint32_t pa = smask(a0);
                const uint32_t pb = (uint32_t)((int32_t)dpp32<DPP_BC2>(base.y) >> 31);
                const uint32_t mk = m2 ? pb : pa, rm = pb & ~m2;
                // w = lanes 0,1: TG = Ha0^Ha1^(pb ? R : 0) | lanes 2,3: Hb0^Hb1 (= TE ^ a0)
                const uint4 w = xand4(p, R, rm);
                const uint4 tab = xand4(w, a0, m2);  // lanes 0,1: TG | lanes 2,3: TE = Hb0^Hb1^a0
                // v = lanes 0,1: WG0 = Ha0 ^ (pa ? TG : 0) | lanes 2,3: WE0 = Hb0 ^ (pb ? TE^a0 : 0)
                const uint4 v = xand4(h, w, mk);
                if (!(q & 1)) row((q & 2) ? TI : 0) = tab;
                put(quad_xor<DPP_XOR2>(v));
            } else if (!HAS_OR || hp.kind == 3) {  // garble.go:446-474
                const uint4 p = quad_xor<DPP_XOR1>(h);              // E0 ^ E1
                if (q == 0) row(0) = lxor(p, R);
                put(lxor(h, sel4(lbit_s(base), p, R)));  // S(a0) ? E1 = E0 ^ (E0^E1) : E0 ^ R
            } else {  // OR: garble.go:412-444
                const uint32_t pa = (base.x >> 31) ^ ((q >> 1) & 1), pb = (base.y >> 31) ^ (q & 1);
                const uint32_t l0 = 2 * pa + pb;
                const uint4 x1 = dpp128<DPP_XOR1>(h), x2 = dpp128<DPP_XOR2>(h), x3 = dpp128<DPP_XOR3>(h);
                const uint4 tk = l0 == 0 ? h : l0 == 1 ? x1 : l0 == 2 ? x2 : x3;  // table[q] = e[q ^ l0]
                const uint4 t0v = dpp128<DPP_BC0>(tk);
                const uint32_t m0 = l0 == 0 ? ~0u : 0u;
                const uint4 c0 = lxor(t0v, land(R, ~m0)), c1 = lxor(t0v, land(R, m0));
                if (q != 0) row((q - 1) << ti_log2) = lxor(tk, q == l0 ? c0 : c1);
                put(c0);
            }
        }
            if (hs.tail) garble_hash_narrow<NR>(buf, u, a, ti_log2, tim, wl, rl, Tt, Wt, lo, hs);
        }
        GC_FPROF(1)
        // Prefetch of the next unit's image and of the header after it, issued AFTER the hash part: vector-memory
        // waits are in order, and the hash part's own waits (the evaluator's table rows; register re-use hazards the
        // compiler guards with vmcnt(0)) would otherwise stall on these loads right after they were issued.  They land
        // during the barrier and the XOR part.
        const FUnit unn_v = load_unit(a.units, ui + 2, vz);
        // image of unit ui + 2 (images are contiguous: it starts where unit ui + 1's ends); past the image's end: the next
        // image or padding
        const uint4 pre_next = a.prog[un.off16 + un.n16 + threadIdx.x];
        if (nh && u.nout) lds_barrier();
        GC_FPROF(2)
        if (u.nout) xor_part<true>(buf, u, a.ogslot, wl, rl, Wt, ti_log2, tim);
        GC_FPROF(4)
        if (threadIdx.x < un.n16) stage[((ui + 1) & 1u) * ustride + threadIdx.x] = pre_cur;
        GC_FPROF(3)
        lds_barrier();
        GC_FPROF(5)
        u = uniform_unit(un);
        un = unn_v;
        pre_cur = pre_next;
    }
    GC_FPROF_EPILOGUE()
    if constexpr (MULTI) {
        // Set through out[] (stream_garble.go:143-157): the job's output labels go back into the wire store.  They were
        // written to W by other waves of this workgroup: the barrier (with its vmcnt(0)) makes them visible.
        __syncthreads();
        const DfBlock *dfb = (const DfBlock *)a.prof;
        for (uint32_t k = threadIdx.x; k < a.nout; k += TF) {
            const uint32_t idx = dfb ? fresh_u32(a.out_idx + k) : a.out_idx[k];
            if (idx == 0xffffffffu) continue;
            if (dfb) {
                // dataflow across launches: the wire must have reached the version this write follows and every read of
                // that version must have happened; the new version goes up behind the label (its store acknowledged first)
                const uint32_t *req = (const uint32_t *)(dfb + 1) + fresh_u32(&dfb->nin) + 3 * k;
                df_wait_reached(dfb->ver + idx, fresh_u32(req), dfb->host_err);
                df_wait_reached(dfb->rd + idx, fresh_u32(req + 1), dfb->host_err);
                store_put(a.store + idx, Wt[a.out_slots[k]]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(dfb->ver + idx, fresh_u32(req + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                store_put(a.store + idx, Wt[a.out_slots[k]]);
            }
        }
    }
}

// this workgroup's job record, word by word through v_readfirstlane: wave-uniform values the compiler keeps in SGPRs
// (a plain struct copy from global memory lands in VGPRs — the kernel has stores it cannot prove disjoint — and with
// them the 128-VGPR budget of a 1024-thread workgroup spills)
__device__ __forceinline__ FlArgs load_job(const FlArgs *job) {
    constexpr int kWords = sizeof(FlArgs) / 4;
    static_assert(sizeof(FlArgs) % 4 == 0, "FlatJob is a whole number of dwords");
    const uint32_t *p = (const uint32_t *)job;
    uint32_t w[kWords];
#pragma unroll
    for (int i = 0; i < kWords; i++) w[i] = __builtin_amdgcn_readfirstlane(p[i]);
    FlArgs a;
    __builtin_memcpy(&a, w, sizeof a);
    return a;
}

template <int NR, bool PROF, bool HAS_OR>
__global__ __launch_bounds__(TF) void k_garble_flat(FlArgs a) {
    garble_flat_body<NR, PROF, HAS_OR, false>(a);
}
// step groups of the streaming engine: workgroup u = launch unit u — one job (a whole, independent one-instance circuit: a
// step, or a chain of steps on its merged plan).  CHAIN: some unit of the launch is a chain of dependent steps that has no
// merged plan (yet): the workgroup runs its jobs one after the other (records first[u] .. first[u] + jobs[first[u]].pad_; each
// job's outputs go back into the wire store before the next one gathers its inputs from there).  A build of its own: the
// loop's state costs the registers the single-job build just fits into (0 against 14 spilled VGPRs).
// Units that depend on one another inside ONE launch (kernels.h: d_sync).  unit_enter: this workgroup's unit = the next ticket;
// then every thread polls one of the done-flags the unit waits for (a bounded wait: ~2 s, then the pinned error word goes up
// and the unit runs anyway — garbage, reported by gc_ctx_coop_check, rather than a hung queue).
__device__ __forceinline__ uint32_t unit_enter(uint32_t *sync) {
    extern __shared__ uint4 smem[];
    uint32_t *scratch = (uint32_t *)smem;  // (the AES table's place: loaded after the barriers below)
    if (threadIdx.x == 0) scratch[0] = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t unit = __builtin_amdgcn_readfirstlane(scratch[0]);
    const uint32_t nun_raw = __builtin_amdgcn_readfirstlane(sync[1]), nun = nun_raw & 0x7fffffffu;
    const uint32_t *off = sync + kSyncHead + nun, *list = off + nun + 1;
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(off[unit]), d1 = __builtin_amdgcn_readfirstlane(off[unit + 1]);
    bool late = false;
    for (uint32_t i = d0 + threadIdx.x; i < d1; i += TF) {
        const uint32_t *flag = sync + kSyncHead + list[i];
        for (uint32_t spins = 0; __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > kDepSpins) {
                late = true;
                break;
            }
        }
    }
    if (late) {
        uint32_t *host_err = *(uint32_t **)(sync + 2);
        if (host_err) __hip_atomic_fetch_max(host_err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    // what those units stored is what this one loads (store_get / store_put, themselves agent-scope accesses past the L1): the
    // flags were read relaxed inside the poll loop; ONE acquire fence at agent scope behind it orders every later load after them
    // (round 6, ADVICE r5: until then the order rested on gfx9's in-order acknowledgement of stores alone)
    (void)nun_raw;
    if (d1 != d0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return unit;
}
__device__ __forceinline__ void unit_leave(uint32_t *sync, uint32_t unit) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every thread's stores into the wire store have been acknowledged
    __syncthreads();
    // the done-flag is a RELEASE at agent scope: everything this workgroup stored happens-before what a unit that sees the flag loads
    if (threadIdx.x == 0) __hip_atomic_store(sync + kSyncHead + unit, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// the same for a record that came up while this kernel was running (the pool: fresh_u32)
__device__ __forceinline__ FlArgs load_job_fresh(const FlArgs *job) {
    constexpr int kWords = sizeof(FlArgs) / 4;
    const uint32_t *p = (const uint32_t *)job;
    uint32_t w[kWords];
#pragma unroll
    for (int i = 0; i < kWords; i++) w[i] = __builtin_amdgcn_readfirstlane(fresh_u32(p + i));
    FlArgs a;
    __builtin_memcpy(&a, w, sizeof a);
    return a;
}

template <int NR, bool HAS_OR, bool CHAIN>
__global__ __launch_bounds__(TF) void k_garble_flat_jobs(const FlArgs *jobs, const uint32_t *first, uint32_t *sync) {
    if constexpr (!CHAIN) {
        const FlArgs a = load_job(jobs + blockIdx.x);
        garble_flat_body<NR, false, HAS_OR, true>(a);
    } else {
        const uint32_t unit = sync ? unit_enter(sync) : blockIdx.x;
        const uint32_t j0 = __builtin_amdgcn_readfirstlane(first[unit]);
        const uint32_t more = __builtin_amdgcn_readfirstlane(jobs[j0].pad_);
        for (uint32_t j = j0; j <= j0 + more; j++) {
            const FlArgs a = load_job(jobs + j);
            garble_flat_body<NR, false, HAS_OR, true>(a);
            if (j != j0 + more) {  // the next job of the chain reads what this one has just stored
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (store_put / store_get: no cache to flush)
                __syncthreads();
            }
        }
        if (sync) unit_leave(sync, unit);
    }
}

// Out-of-order issue (kernels.h: PoolCtl): a workgroup claims the lowest published unit nobody has claimed, runs its records,
// counts it done, and claims again; it leaves when nothing is published that is not claimed.  (HAS_OR always: a workgroup of an
// earlier launch may claim a unit of a later one.)
template <int NR>
__global__ __launch_bounds__(TF) void k_garble_flat_pool(PoolCtl *ctl) {
    for (;;) {
        extern __shared__ uint4 smem[];
        uint32_t *scratch = (uint32_t *)smem;  // (the AES table's place: a unit's first record loads the table behind the barriers below)
        if (threadIdx.x == 0) {
            uint32_t t = 0xffffffffu;
            const uint32_t persist = __hip_atomic_load(&ctl->persist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool seen_stop = false;
            for (uint32_t idle = 0;;) {
                uint32_t h = __hip_atomic_load(&ctl->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t tl = __hip_atomic_load(&ctl->tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if ((int32_t)(tl - h) > 0) {
                    if (__hip_atomic_compare_exchange_strong(&ctl->head, &h, h + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        t = h;
                        break;
                    }
                    continue;
                }
                if (!persist) break;
                // persistent: nothing to claim now.  `stop` goes up BEHIND the host's last publication: once it is seen, one more
                // look at the tail is final.  (Bounded all the same: ~20 s of idle polls.)
                if (__hip_atomic_load(&ctl->stop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (seen_stop) break;
                    seen_stop = true;
                    continue;
                }
                __builtin_amdgcn_s_sleep(32);
                if (++idle > (1u << 24)) break;
            }
            scratch[0] = t;
        }
        __syncthreads();
        const uint32_t t = __builtin_amdgcn_readfirstlane(scratch[0]);
        __syncthreads();
        if (t == 0xffffffffu) return;
        // The unit's records, wire maps and dataflow block came up by DMA after this workgroup's KERNEL began (that is the point:
        // it runs units of later launches), into upload regions that earlier launches used too: this XCD's L2 — coherent with
        // the others only at kernel boundaries — may still hold the old lines.  An acquire at agent scope drops them (every wave:
        // its vector L1 and scalar cache as well).  (Found as faults on null / stale pointers once the workgroups looped.)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const PoolEntry *e = ctl->ring + (t & (kPoolRing - 1u));
        const uint64_t recp = __hip_atomic_load((const uint64_t *)&e->rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t mg = __hip_atomic_load((const uint64_t *)&e->more, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (readfirstlane returns a signed int: the low half must not sign-extend into the high one)
        const FlArgs *rec = (const FlArgs *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(recp >> 32)) << 32) |
                                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)recp));
        const uint32_t more = __builtin_amdgcn_readfirstlane((uint32_t)mg), group = __builtin_amdgcn_readfirstlane((uint32_t)(mg >> 32));
        for (uint32_t j = 0; j <= more; j++) {
            const FlArgs a = load_job_fresh(rec + j);
            garble_flat_body<NR, false, true, true>(a);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (store_put / store_get: no cache to flush)
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctl->done + group, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&ctl->done_total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (__builtin_amdgcn_readfirstlane(ctl->pad_)) return;  // (developer aid, GC_DF_DEBUG=2: one unit per workgroup)
    }
}
// the units [first_ticket, first_ticket + n) of a launch go into the ring; the tail moves behind them
__global__ void k_pool_publish(PoolCtl *ctl, const PoolEntry *entries, uint32_t first_ticket, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const PoolEntry e = entries[i];
        PoolEntry *d = ctl->ring + ((first_ticket + i) & (kPoolRing - 1u));
        __hip_atomic_store((uint64_t *)&d->rec, (uint64_t)(uintptr_t)e.rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((uint64_t *)&d->more, (uint64_t)e.more | ((uint64_t)e.group << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&ctl->tail, first_ticket + n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// one thread waits until a counter has reached `want` (the units of a group / everything published so far are done): what stands
// behind it on its stream — the group's serialiser, a big step's pass — sees their labels and table rows
__global__ void k_pool_wait(const uint32_t *ctr, uint32_t want, uint32_t *host_err) {
    for (uint32_t spins = 0; (int32_t)(__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want) < 0;) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > (1u << 24)) {  // (~20 s: a unit may be a 2 ms multiplier behind thirty others)
            if (host_err) __hip_atomic_fetch_max(host_err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}

__global__ void k_pool_stop(PoolCtl *ctl) { __hip_atomic_store(&ctl->stop, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

template <int NR, bool PROF, bool HAS_OR, bool MULTI>
__device__ __forceinline__ void eval_flat_body(const FlArgs &a) {
    GC_FL_PROLOGUE(false)
    const uint4 *Tt = a.T + (size_t)tile * a.t_tile;
    for (uint32_t ui = 0; ui < a.nunits; ui++) {
        const uint4 *buf = stage + (ui & 1u) * ustride;
        const uint32_t nh = u.n_and + u.n_or + u.n_inv;
        GC_FPROF(0)
        const uint32_t e_hash = hlanes<1, 0, 0>(u, ti_log2);
        if (nh && u.n_or == 0 && e_hash <= kNarrowLanes) {  // a narrow unit: column-sliced as a whole
            eval_hash_narrow<NR, PROF>(buf, u, a, ti_log2, tim, wl, Tt, Wt, lo, HashSplit{0, e_hash, 0}, pacc, &plast);
        } else if (nh) {
            // wide form for whole sets of four waves, column-sliced form for what is left of the last pass
            const HashSplit hs = split_hash_lanes(e_hash, u.n_or != 0);
            const uint32_t e_all = hs.wide_end;
        for (uint32_t t0 = 0; t0 + wave_base < e_all; t0 += TF) {  // scalar test: a wave without lanes leaves at once
            const HP hp = hpos<1, 0, 0, HAS_OR>(t0 + threadIdx.x, u, ti_log2, tim);
            if (hp.kind == 0) continue;
            const uint4 dv = buf[hp.g];
            const FDesc d{dv.x, dv.y, dv.z, dv.w};
            const uint32_t inst = hp.inst, q = hp.q;
            const uint32_t rowb = (((d.row_op & kRowMask) << ti_log2) + inst) << 4;  // 32-bit byte offset, as in the garbler
            auto row = [&](uint32_t r) -> const uint4 & { return *(const uint4 *)((const char *)Tt + (rowb + (r << 4))); };
            // AND: lane q hashes operand q (a, b) with tweak + q and needs table row q; INV: operand a, row 0
            const uint32_t mq = q ? ~0u : 0u;  // q is 0 for INV / OR lanes
            const uint32_t sh = ti_log2 + 4, ib = (uint32_t)(uintptr_t)wl + (inst << 4);
            const uint4 va = lds_label(ib, half_of(d.lin, !(HAS_OR && hp.kind == 2) && q), sh);
            uint4 x = va, tab = make_uint4(0, 0, 0, 0);
            uint32_t k[4];
            if (!HAS_OR || hp.kind != 2) {
                tab = row(q ? TI : 0);  // issued before the hash: arrives while the AES runs
                whiten_half(x, d.tweak + q, rkr, k);
            } else {  // OR (eval.go:80-94): both operands, row index - 1 (index 0 has no row)
                const uint4 vb = lds_label(ib, d.lin >> 16, sh);
                const uint32_t index = (lbit_s(va) ? 2u : 0u) | (lbit_s(vb) ? 1u : 0u);
                if (index > 0) tab = row((index - 1) << ti_log2);
                whiten_k(va, vb, d.tweak, rkr, k);
            }
            GC_FPROF(6)  // debug profile: slot 6 = operand fetch + key set-up, slot 7 = AES, slot 1 = combine + stores
            const uint4 h = hash_dual_whitened<NR>(k, rkr, te, lo);
            GC_FPROF(7)
            auto put = [&](const uint4 &out_label) {  // inside every branch: no result merge, no register copies
                if (q == 0) {
                    lds_label_put(ib, d.lout & 0xffffu, sh, out_label);
                    if (d.lout & kFStoreGlobal)
                        Wt[((size_t)a.hgslot[u.hfirst + hp.g] << ti_log2) + inst] = out_label;
                }
            };
            if (hp.kind == 1) {  // eval.go:53-78
                const uint4 av = dpp128<DPP_PAIR0>(x);
                // lane 0: WG = H(a) ^ (sa ? TG : 0); lane 1: WE = H(b) ^ (sb ? TE^a : 0)
                const uint4 v = xand4(h, xand4(tab, av, mq), smask(x));
                put(quad_xor<DPP_XOR1>(v));
            } else if (!HAS_OR || hp.kind == 3) {  // eval.go:96-109
                put(xand4(h, tab, smask(x)));
            } else {  // eval.go:80-94 (tab is zero for index 0)
                put(lxor(h, tab));
            }
        }
            if (hs.tail) eval_hash_narrow<NR>(buf, u, a, ti_log2, tim, wl, Tt, Wt, lo, hs);
        }
        GC_FPROF(1)
        // Prefetch of the next unit's image and of the header after it, issued AFTER the hash part: vector-memory
        // waits are in order, and the hash part's own waits (the evaluator's table rows; register re-use hazards the
        // compiler guards with vmcnt(0)) would otherwise stall on these loads right after they were issued.  They land
        // during the barrier and the XOR part.
        const FUnit unn_v = load_unit(a.units, ui + 2, vz);
        // image of unit ui + 2 (images are contiguous: it starts where unit ui + 1's ends); past the image's end: the next
        // image or padding
        const uint4 pre_next = a.prog[un.off16 + un.n16 + threadIdx.x];
        if (nh && u.nout) lds_barrier();
        GC_FPROF(2)
        if (u.nout) xor_part<false>(buf, u, a.ogslot, wl, rl, Wt, ti_log2, tim);
        GC_FPROF(4)
        if (threadIdx.x < un.n16) stage[((ui + 1) & 1u) * ustride + threadIdx.x] = pre_cur;
        GC_FPROF(3)
        lds_barrier();
        GC_FPROF(5)
        u = uniform_unit(un);
        un = unn_v;
        pre_cur = pre_next;
    }
    GC_FPROF_EPILOGUE()
    if constexpr (MULTI) {
        // Set through out[] (stream_garble.go:143-157): the job's output labels go back into the wire store.  They were
        // written to W by other waves of this workgroup: the barrier (with its vmcnt(0)) makes them visible.
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < a.nout; k += TF) {
            const uint32_t idx = a.out_idx[k];
            if (idx != 0xffffffffu) store_put(a.store + idx, Wt[a.out_slots[k]]);
        }
    }
}

template <int NR, bool PROF, bool HAS_OR>
__global__ __launch_bounds__(TF) void k_eval_flat(FlArgs a) {
    eval_flat_body<NR, PROF, HAS_OR, false>(a);
}
template <int NR, bool HAS_OR, bool CHAIN>
__global__ __launch_bounds__(TF) void k_eval_flat_jobs(const FlArgs *jobs, const uint32_t *first, uint32_t *sync) {
    if constexpr (!CHAIN) {
        const FlArgs a = load_job(jobs + blockIdx.x);
        eval_flat_body<NR, false, HAS_OR, true>(a);
    } else {
        const uint32_t unit = sync ? unit_enter(sync) : blockIdx.x;
        const uint32_t j0 = __builtin_amdgcn_readfirstlane(first[unit]);
        const uint32_t more = __builtin_amdgcn_readfirstlane(jobs[j0].pad_);
        for (uint32_t j = j0; j <= j0 + more; j++) {
            const FlArgs a = load_job(jobs + j);
            eval_flat_body<NR, false, HAS_OR, true>(a);
            if (j != j0 + more) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (store_put / store_get: no cache to flush)
                __syncthreads();
            }
        }
        if (sync) unit_leave(sync, unit);
    }
}

size_t fused_flat_bytes(uint32_t nls, uint32_t ti_log2, uint32_t ustride) {
    return (size_t)(kStageOff + 2 * ustride) * sizeof(uint4) + ((size_t)(nls + 1) << ti_log2) * sizeof(uint4);
}

template <typename K>
static hipError_t launch_fl(K kern, const FlArgs &a, uint32_t ntiles, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(TF), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_fused_flat(bool eval, const FusedFlatArgs &f, const BatchGeom &g, hipStream_t s) {
    FlArgs a{};
    a.prog = (const uint4 *)f.prog;
    a.units = f.units;
    a.hgslot = f.hgslot;
    a.ogslot = f.ogslot;
    a.in_lds = f.in_lds;
    a.nunits = f.nunits;
    a.ninputs = f.ninputs;
    a.ti_log2 = g.ti_log2;
    a.zslot = f.nls - 1;
    a.ustride = f.ustride;
    a.w_tile = g.lw.tile_stride;
    a.t_tile = g.lt.tile_stride;
    a.W = f.W;
    a.R = f.R;
    a.T = f.T;
    a.rk = f.rk;
    a.te0 = f.te0;
    a.prof = f.prof;
    a.rnd = eval ? nullptr : f.rnd;
    a.Rout = const_cast<uint4 *>(f.R);
    a.batch = g.batch;
    if (a.nunits == 0) return hipSuccess;
    const size_t lds = fused_flat_bytes(f.nls, g.ti_log2, f.ustride);
#define GC_M3(KERN, NR)                                                                      \
    (f.prof ? (f.has_or ? launch_fl(KERN<NR, true, true>, a, g.ntiles, lds, s)               \
                        : launch_fl(KERN<NR, true, false>, a, g.ntiles, lds, s))             \
            : f.has_or ? launch_fl(KERN<NR, false, true>, a, g.ntiles, lds, s)              \
                       : launch_fl(KERN<NR, false, false>, a, g.ntiles, lds, s))
#define GC_M2(KERN) (f.rounds == 10 ? GC_M3(KERN, 10) : f.rounds == 12 ? GC_M3(KERN, 12) : GC_M3(KERN, 14))
    return eval ? GC_M2(k_eval_flat) : GC_M2(k_garble_flat);
#undef GC_M2
#undef GC_M3
}

template <typename K>
static hipError_t launch_jobs(K kern, const FlatJob *jobs, const uint32_t *first, uint32_t *sync, uint32_t nunits, size_t lds,
                              hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nunits), dim3(TF), lds, s, jobs, first, sync);
    return hipGetLastError();
}

template <typename K>
static hipError_t launch_pool(K kern, PoolCtl *ctl, uint32_t nworkers, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nworkers), dim3(TF), lds, s, ctl);
    return hipGetLastError();
}
hipError_t launch_fused_flat_pool(int rounds, PoolCtl *ctl, uint32_t nworkers, size_t lds_bytes, hipStream_t s) {
    if (nworkers == 0) return hipSuccess;
    return rounds == 10 ? launch_pool(k_garble_flat_pool<10>, ctl, nworkers, lds_bytes, s)
           : rounds == 12 ? launch_pool(k_garble_flat_pool<12>, ctl, nworkers, lds_bytes, s)
                          : launch_pool(k_garble_flat_pool<14>, ctl, nworkers, lds_bytes, s);
}
void launch_pool_publish(PoolCtl *ctl, const PoolEntry *d_entries, uint32_t first_ticket, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pool_publish, dim3(1), dim3(256), 0, s, ctl, d_entries, first_ticket, n);
}
void launch_pool_stop(PoolCtl *ctl, hipStream_t s) { hipLaunchKernelGGL(k_pool_stop, dim3(1), dim3(1), 0, s, ctl); }
void launch_pool_wait(const uint32_t *d_counter, uint32_t want, uint32_t *d_host_err, hipStream_t s) {
    hipLaunchKernelGGL(k_pool_wait, dim3(1), dim3(1), 0, s, d_counter, want, d_host_err);
}

hipError_t launch_fused_flat_jobs(bool eval, int rounds, bool has_or, const FlatJob *d_jobs, const uint32_t *d_first, uint32_t nunits,
                                  size_t lds_bytes, hipStream_t s, uint32_t *d_sync) {
    if (nunits == 0) return hipSuccess;
    if (d_sync && !d_first) return hipErrorInvalidValue;
    const bool chain = d_first != nullptr;  // (nullptr: unit u is record u)
#define GC_J4(KERN, NR, OR) (chain ? launch_jobs(KERN<NR, OR, true>, d_jobs, d_first, d_sync, nunits, lds_bytes, s) : launch_jobs(KERN<NR, OR, false>, d_jobs, d_first, nullptr, nunits, lds_bytes, s))
#define GC_J3(KERN, NR) (has_or ? GC_J4(KERN, NR, true) : GC_J4(KERN, NR, false))
#define GC_J2(KERN) (rounds == 10 ? GC_J3(KERN, 10) : rounds == 12 ? GC_J3(KERN, 12) : GC_J3(KERN, 14))
    return eval ? GC_J2(k_eval_flat_jobs) : GC_J2(k_garble_flat_jobs);
#undef GC_J2
#undef GC_J3
#undef GC_J4
}

}  // namespace gc
