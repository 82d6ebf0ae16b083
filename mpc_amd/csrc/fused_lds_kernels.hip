// fused_lds_kernels.hip — fused schedule with LDS-RESIDENT wire labels (the production path).
//
// One launch per Garble / Eval pass.  A workgroup owns a tile of TI instances and walks the whole
// circuit itself; tiles are independent (no grid barrier, no inter-workgroup traffic).
//   * The live wire labels of the tile never leave the CU: they sit in LDS slots that the host plan
//     recycles after the last reader (aes_128: <= ~1.03k live labels -> 16.5 KiB per instance).  The
//     HBM traffic of a pass is: input labels in, garbled tables out (garble) / in (eval), output labels
//     out — nothing else.
//   * Steps follow the hash-phase schedule (plan.h): all table-producing gates of one non-free depth
//     are hashed together (89 phases for aes_128); the XOR sub-levels in between touch LDS only.
//   * The inter-step barrier drains LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): table stores
//     and prefetches stay in flight across it.
//   * Descriptors never stall a step: the plan cuts the schedule into chunks (one hash phase + its XOR
//     sub-levels); while chunk c executes, the threads prefetch chunk c+1's descriptors into registers
//     and drop them into an LDS staging area at the chunk boundary.
//   * Few waves, deep ILP: per-step bookkeeping is paid per wave, so the workgroup is small (THREADS)
//     and every hash lane runs ILP independent AES chains instead.
// Hash work decomposition (lanes of a hash step):
//   garble: AND 4 lanes per (gate, instance): H(a0), H(a1), H(b0), H(b1)   (circuit/garble.go:362-376)
//           OR  4 lanes: the four enc(a_u, b_v, 0, id)                      (garble.go:421-424)
//           INV 2 lanes: enc(a0), enc(a1)                                   (garble.go:453-454)
//   eval:   AND 2 lanes: H(a, j0), H(b, j1);  OR 1 lane;  INV 1 lane       (circuit/eval.go:68-72,93,108)
// The 2-4 lanes of a gate-instance exchange their hashes with DPP quad permutes (no LDS traffic).
// LDS map (dynamic): [0, 64 KiB) perm-addressed dual AES table | 16 KiB descriptor stage | 2 KiB step
// stage | R of the tile | wire slots [slot][TI].
#include <cstdlib>

#include "aes_device.h"
#include "kernels.h"

namespace gc {

constexpr int DPP_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;   // [2,3,0,1]
constexpr int DPP_XOR3 = 0x1B;   // [3,2,1,0]
constexpr int DPP_BC0 = 0x00;    // [0,0,0,0]
constexpr int DPP_BC2 = 0xAA;    // [2,2,2,2]
constexpr int DPP_PAIR0 = 0xA0;  // [0,0,2,2]

template <int CTRL>
__device__ __forceinline__ uint32_t ldpp32(uint32_t v) {
    // quad permutes have no invalid source lanes inside a full quad: no "old" value, so no register initialisation
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ uint4 ldpp128(uint4 v) {
    return make_uint4(ldpp32<CTRL>(v.x), ldpp32<CTRL>(v.y), ldpp32<CTRL>(v.z), ldpp32<CTRL>(v.w));
}

// workgroup barrier that only drains LDS traffic (global stores / prefetches keep flying)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define GC_LPROF(slot)                                               \
    if constexpr (PROF) {                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
        const uint64_t now__ = __builtin_amdgcn_s_memtime();         \
        pacc[slot] += now__ - plast;                                 \
        plast = now__;                                               \
    }

// staging area, in uint4 units: TWO buffers of [kChunkDescs descriptors | kChunkSteps steps (32 B each)];
// chunk c reads buffer c&1 while the prefetched chunk c+1 is committed into the other one
constexpr uint32_t kStageBuf = kChunkDescs + kChunkSteps * 2;
constexpr uint32_t kStageOff = kTeDualBytes / 16;
constexpr uint32_t kStageEnd = kStageOff + 2 * kStageBuf;

__device__ __forceinline__ Step read_step(const uint4 *stage, uint32_t s) {
    const uint4 lo = stage[2 * s], hi = stage[2 * s + 1];
    Step st;
    st.first = __builtin_amdgcn_readfirstlane(lo.x);
    st.count = __builtin_amdgcn_readfirstlane(lo.y);
    st.nonfree = __builtin_amdgcn_readfirstlane(lo.z);
    st.n_and = __builtin_amdgcn_readfirstlane(lo.w);
    st.n_or = __builtin_amdgcn_readfirstlane(hi.x);
    st.n_inv = __builtin_amdgcn_readfirstlane(hi.y);
    return st;
}

__device__ __forceinline__ Chunk read_chunk(const Chunk *chunks, uint32_t c, uint32_t n) {
    Chunk ch{0, 0, 0, 0};
    if (c < n) {
        ch.first_step = __builtin_amdgcn_readfirstlane(chunks[c].first_step);
        ch.nsteps = __builtin_amdgcn_readfirstlane(chunks[c].nsteps);
        ch.first_desc = __builtin_amdgcn_readfirstlane(chunks[c].first_desc);
        ch.ndesc = __builtin_amdgcn_readfirstlane(chunks[c].ndesc);
    }
    return ch;
}

struct LdsArgs {
    const FDesc *descs;
    const uint32_t *gslot;
    const Step *steps;
    const Chunk *chunks;
    const uint16_t *in_lds;
    uint32_t nsteps, nchunks, ninputs, nls, ti_log2;
    size_t w_tile, t_tile;
    uint4 *W;
    const uint4 *R;
    uint4 *T;
    const uint32_t *rk;
    const uint32_t *te0;
    uint64_t *prof;
};

// hash-step lane -> (kind, gate, instance, sub-lane); kinds: 0 none, 1 AND, 2 OR, 3 INV
struct HPos {
    uint32_t kind, g, inst, q;
};
template <int LQA, int LQO, int LQI>
__device__ __forceinline__ HPos hclassify(uint32_t t, uint32_t e_and, uint32_t e_or, uint32_t e_all, uint32_t n_and,
                                          uint32_t n_or, uint32_t ti_log2, uint32_t tim) {
    HPos p{0, 0, 0, 0};
    if (t < e_and) {
        p.kind = 1;
        p.g = t >> (ti_log2 + LQA);
        p.inst = (t >> LQA) & tim;
        p.q = t & ((1u << LQA) - 1);
    } else if (t < e_or) {
        const uint32_t u = t - e_and;
        p.kind = 2;
        p.g = n_and + (u >> (ti_log2 + LQO));
        p.inst = (u >> LQO) & tim;
        p.q = u & ((1u << LQO) - 1);
    } else if (t < e_all) {
        const uint32_t u = t - e_or;
        p.kind = 3;
        p.g = n_and + n_or + (u >> (ti_log2 + LQI));
        p.inst = (u >> LQI) & tim;
        p.q = u & ((1u << LQI) - 1);
    }
    return p;
}

__device__ __forceinline__ FDesc stage_desc(const uint4 *stage_d, const FDesc *descs, bool direct, uint32_t rel,
                                            uint32_t first, uint32_t g) {
    if (direct) return descs[first + g];
    const uint4 dv = stage_d[rel + g];
    return FDesc{dv.x, dv.y, dv.z, dv.w};
}

// common prologue / chunk plumbing of both kernels
#define GC_LDS_PROLOGUE(LOAD_R)                                                                                \
    extern __shared__ uint4 smem[];                                                                            \
    uint32_t *te = (uint32_t *)smem;                                                                           \
    const uint32_t ti_log2 = a.ti_log2, TI = 1u << ti_log2, tim = TI - 1;                                      \
    uint4 *stage_d = smem + kStageOff;                                                                         \
    uint4 *stage_s = stage_d + kChunkDescs;                                                                    \
    uint4 *rl = smem + kStageEnd;                                                                              \
    uint4 *wl = rl + TI;                                                                                       \
    load_te_dual(te, a.te0);                                                                                   \
    uint32_t rkr[4 * (NR + 1)];                                                                                \
    load_round_keys<NR>(rkr, a.rk);                                                                            \
    uint4 *Wt = a.W + (size_t)blockIdx.x * a.w_tile;                                                           \
    if (LOAD_R && threadIdx.x < TI) rl[threadIdx.x] = a.R[(size_t)blockIdx.x * TI + threadIdx.x];             \
    for (uint32_t i = threadIdx.x; i < (a.ninputs << ti_log2); i += THREADS) {                                 \
        const uint32_t w = i >> ti_log2, ls = a.in_lds[w];                                                     \
        if (ls != 0xffffu) wl[(ls << ti_log2) + (i & tim)] = Wt[i]; /* global input slots are [w][TI] */      \
    }                                                                                                          \
    Chunk ch = read_chunk(a.chunks, 0, a.nchunks);                                                             \
    if (ch.ndesc <= kChunkDescs)                                                                               \
        for (uint32_t i = threadIdx.x; i < ch.ndesc; i += THREADS)                                             \
            stage_d[i] = ((const uint4 *)a.descs)[ch.first_desc + i];                                          \
    if (threadIdx.x < ch.nsteps) {                                                                             \
        const Step sv = a.steps[ch.first_step + threadIdx.x];                                                  \
        stage_s[2 * threadIdx.x] = make_uint4(sv.first, sv.count, sv.nonfree, sv.n_and);                       \
        stage_s[2 * threadIdx.x + 1] = make_uint4(sv.n_or, sv.n_inv, 0, 0);                                    \
    }                                                                                                          \
    __syncthreads();                                                                                           \
    const uint32_t lo = te_lane_off();                                                                         \
    /* XOR sub-levels: NW worker waves, each owning OI = TI / NW instances of the tile */                      \
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;           \
    const uint32_t nw_log2 = ti_log2 < (uint32_t)__builtin_ctz(THREADS / 64) ? ti_log2                         \
                                                                             : (uint32_t)__builtin_ctz(THREADS / 64); \
    const uint32_t NW = 1u << nw_log2, oi_log2 = ti_log2 - nw_log2;                                            \
    const uint32_t xinst = (wave << oi_log2) + (lane & ((1u << oi_log2) - 1));                                 \
    uint64_t pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = 0;                                                                \
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();

#define GC_CHUNK_PREFETCH()                                                                                    \
    stage_d = smem + kStageOff + (c & 1u) * kStageBuf;                                                         \
    stage_s = stage_d + kChunkDescs;                                                                           \
    uint4 *stage_dn = smem + kStageOff + ((c + 1) & 1u) * kStageBuf, *stage_sn = stage_dn + kChunkDescs;       \
    const Chunk nx = read_chunk(a.chunks, c + 1, a.nchunks);                                                   \
    const bool stage_next = nx.ndesc <= kChunkDescs;                                                           \
    uint4 pre_d[PF];                                                                                           \
    _Pragma("unroll") for (int i = 0; i < PF; i++) {                                                           \
        const uint32_t kk = threadIdx.x + i * THREADS;                                                         \
        pre_d[i] = (stage_next && kk < nx.ndesc) ? ((const uint4 *)a.descs)[nx.first_desc + kk]                \
                                                 : make_uint4(0, 0, 0, 0);                                     \
    }                                                                                                          \
    Step pre_s{0, 0, 0, 0, 0, 0};                                                                              \
    if (threadIdx.x < nx.nsteps) pre_s = a.steps[nx.first_step + threadIdx.x];                                 \
    const bool direct = ch.ndesc > kChunkDescs;

#define GC_CHUNK_COMMIT()                                                                                      \
    /* the other staging buffer was last read by chunk c-1, which ended with a barrier: safe to fill now */  \
    _Pragma("unroll") for (int i = 0; i < PF; i++) {                                                           \
        const uint32_t kk = threadIdx.x + i * THREADS;                                                         \
        if (stage_next && kk < nx.ndesc) stage_dn[kk] = pre_d[i];                                              \
    }                                                                                                          \
    if (threadIdx.x < nx.nsteps) {                                                                             \
        stage_sn[2 * threadIdx.x] = make_uint4(pre_s.first, pre_s.count, pre_s.nonfree, pre_s.n_and);          \
        stage_sn[2 * threadIdx.x + 1] = make_uint4(pre_s.n_or, pre_s.n_inv, 0, 0);                            \
    }

#define GC_PROF_EPILOGUE()                                                                                     \
    if constexpr (PROF) {                                                                                      \
        if (threadIdx.x == 0 || threadIdx.x == 192)                                                   \
            for (int i = 0; i < 8; i++) a.prof[(size_t)blockIdx.x * 16 + (threadIdx.x ? 8 : 0) + i] = pacc[i]; \
    }

// ------------------------------------------------------------------------------------------------------
template <int NR, int THREADS, int ILP, bool STORE_ALL, bool PROF>
__global__ __launch_bounds__(THREADS) void k_garble_lds(LdsArgs a) {
    constexpr int PF = ((int)kChunkDescs + THREADS - 1) / THREADS;
    GC_LDS_PROLOGUE(true)
    uint4 *Tt = a.T + (size_t)blockIdx.x * a.t_tile;

    for (uint32_t c = 0; c < a.nchunks; c++) {
        GC_CHUNK_PREFETCH()
        // a chunk is [hash phase]? followed by XOR sub-levels (plan.cpp cuts before every hash phase)
        uint32_t sidx = 0, xor_rel0 = 0;
        {
            const Step st = read_step(stage_s, 0);
            const uint32_t rel = st.first - ch.first_desc;
            GC_LPROF(0)
            if (st.nonfree != 0) {
                sidx = 1;
                xor_rel0 = st.count;
                // ---- hash phase ----
                const uint32_t e_and = (st.n_and << ti_log2) << 2;
                const uint32_t e_or = e_and + ((st.n_or << ti_log2) << 2);
                const uint32_t e_all = e_or + ((st.n_inv << ti_log2) << 1);
                for (uint32_t t0 = 0; t0 < e_all; t0 += THREADS * ILP) {
                    HPos hp[ILP];
                    FDesc d[ILP];
                    uint4 base[ILP];
                    uint32_t k[ILP][4];
#pragma unroll
                    for (int j = 0; j < ILP; j++) {
                        hp[j] = hclassify<2, 2, 1>(t0 + j * THREADS + threadIdx.x, e_and, e_or, e_all, st.n_and,
                                                   st.n_or, ti_log2, tim);
                        k[j][0] = k[j][1] = k[j][2] = k[j][3] = 0;
                        base[j] = make_uint4(0, 0, 0, 0);
                        d[j] = FDesc{0, 0, 0, 0};
                        if (hp[j].kind) {
                            d[j] = stage_desc(stage_d, a.descs, direct, rel, st.first, hp[j].g);
                            const uint32_t inst = hp[j].inst, q = hp[j].q;
                            const uint4 R = rl[inst];
                            const uint4 va = wl[((d[j].lin & 0xffffu) << ti_log2) + inst];
                            if (hp[j].kind == 2) {  // OR: e[2u+v] = enc(a_u, b_v, 0, id)  (garble.go:74-83)
                                const uint4 vb = wl[((d[j].lin >> 16) << ti_log2) + inst];
                                const uint4 x = lxor(va, land(R, (q & 2) ? ~0u : 0u));
                                const uint4 y = lxor(vb, land(R, (q & 1) ? ~0u : 0u));
                                base[j] = make_uint4(x.y, y.y, 0, 0);
                                make_k(x, y, d[j].tweak, k[j]);
                            } else {  // AND q=0..3 -> a0,a1,b0,b1 ; INV q=0,1 -> a0,a1 ; K = 2x ^ tweak
                                const bool second = (hp[j].kind == 1) && (q & 2);
                                base[j] = second ? wl[((d[j].lin >> 16) << ti_log2) + inst] : va;
                                const uint4 x = lxor(base[j], land(R, (q & 1) ? ~0u : 0u));
                                make_k_half(x, d[j].tweak + (second ? 1u : 0u), k[j]);
                            }
                        }
                    }
                    uint4 h[ILP];
                    if constexpr (ILP == 2) {
                        if (__any(hp[1].kind != 0)) {
                            hash_dual_n<NR, 2>(k, h, rkr, te, lo);
                        } else {  // this wave only has first-lane work: single AES chain
                            h[0] = hash_dual<NR>(k[0], rkr, te, lo);
                            h[1] = make_uint4(0, 0, 0, 0);
                        }
                    } else {
                        hash_dual_n<NR, ILP>(k, h, rkr, te, lo);
                    }
#pragma unroll
                    for (int j = 0; j < ILP; j++) {
                        const uint32_t kind = hp[j].kind, inst = hp[j].inst, q = hp[j].q;
                        if (kind == 0) continue;
                        const uint4 R = rl[inst];
                        uint4 *row = Tt + ((size_t)(d[j].row_op & kRowMask) << ti_log2) + inst;
                        uint4 out_label;
                        if (kind == 1) {  // garble.go:353-395
                            const uint4 p = lxor(h[j], ldpp128<DPP_XOR1>(h[j]));  // 0,1: Ha0^Ha1 ; 2,3: Hb0^Hb1
                            const uint4 a0 = ldpp128<DPP_BC0>(base[j]);
                            const uint32_t pa = smask(a0);
                            const uint32_t pb = (uint32_t)((int32_t)ldpp32<DPP_BC2>(base[j].y) >> 31);
                            uint4 v, tab;
                            if (q & 2) {
                                tab = lxor(p, a0);                        // TE = Hb0^Hb1^a0
                                v = lxor(h[j], land(lxor(tab, a0), pb));  // WE0 = Hb0 ^ (pb ? TE^a0 : 0)
                            } else {
                                tab = lxor(p, land(R, pb));               // TG = Ha0^Ha1^(pb?R:0)
                                v = lxor(h[j], land(tab, pa));            // WG0 = Ha0 ^ (pa ? TG : 0)
                            }
                            out_label = lxor(v, ldpp128<DPP_XOR2>(v));
                            if (q == 0) row[0] = tab;
                            else if (q == 2) row[TI] = tab;
                        } else if (kind == 3) {  // garble.go:446-474
                            const uint4 p = lxor(h[j], ldpp128<DPP_XOR1>(h[j]));          // E0 ^ E1
                            out_label = lbit_s(base[j]) ? lxor(p, h[j]) : lxor(h[j], R);  // S(a0) ? E1 : E0^R
                            if (q == 0) row[0] = lxor(p, R);
                        } else {  // OR: garble.go:412-444
                            const uint32_t pa = (base[j].x >> 31) ^ ((q >> 1) & 1), pb = (base[j].y >> 31) ^ (q & 1);
                            const uint32_t l0 = 2 * pa + pb;
                            const uint4 x1 = ldpp128<DPP_XOR1>(h[j]), x2 = ldpp128<DPP_XOR2>(h[j]),
                                        x3 = ldpp128<DPP_XOR3>(h[j]);
                            const uint4 tk = l0 == 0 ? h[j] : l0 == 1 ? x1 : l0 == 2 ? x2 : x3;  // table[q] = e[q^l0]
                            const uint4 t0v = ldpp128<DPP_BC0>(tk);
                            const uint32_t m0 = l0 == 0 ? ~0u : 0u;
                            const uint4 c0 = lxor(t0v, land(R, ~m0)), c1 = lxor(t0v, land(R, m0));
                            out_label = c0;
                            if (q != 0) row[(size_t)(q - 1) << ti_log2] = lxor(tk, q == l0 ? c0 : c1);
                        }
                        if (q == 0) {
                            wl[((d[j].lout & 0xffffu) << ti_log2) + inst] = out_label;
                            if (STORE_ALL || (d[j].lout & kFStoreGlobal))
                                Wt[((size_t)a.gslot[st.first + hp[j].g] << ti_log2) + inst] = out_label;
                        }
                    }
                }
                GC_LPROF(1)
                if (sidx < ch.nsteps) {
                    lds_barrier();
                    GC_LPROF(2)
                }
            }
        }
        // ---- XOR sub-levels (garble.go:331-351): LDS in, LDS out, NO workgroup barriers.  Wave w owns
        // the instances [w*OI, (w+1)*OI) of the tile; a wave's DS operations execute in order, so the
        // labels one sub-level writes are visible to the next one without any synchronisation.
        GC_CHUNK_COMMIT()
        GC_LPROF(3)
        if (sidx < ch.nsteps) {
            if (wave < NW && !direct) {
                // flat stream over all XOR descriptors of the chunk: an iteration takes up to 64>>oi gates
                // and stops at the next sub-level start (kFLevelStart); the next descriptors are fetched
                // from the LDS stage before the current labels are touched
                const uint32_t gl = lane >> oi_log2;
                uint32_t p = xor_rel0;
                const uint32_t end = ch.ndesc;
                uint4 dv = p + gl < end ? stage_d[p + gl] : make_uint4(0, 0, 0, 0);
                while (p < end) {
                    // lanes whose descriptor belongs to the sub-level of the first lane take part; the label
                    // reads are issued first, the bookkeeping for the next iteration overlaps their latency
                    const uint32_t lvl = __builtin_amdgcn_readfirstlane(dv.z);
                    const bool act = p + gl < end && dv.z == lvl;
                    uint4 va = make_uint4(0, 0, 0, 0), vb = va;
                    if (act) {
                        va = wl[((dv.x & 0xffffu) << ti_log2) + xinst];
                        vb = wl[((dv.x >> 16) << ti_log2) + xinst];
                    }
                    const uint32_t n = (uint32_t)__builtin_popcountll(__ballot(act)) >> oi_log2;
                    const uint4 dn = p + n + gl < end ? stage_d[p + n + gl] : make_uint4(0, 0, 0, 0);
                    if (act) {
                        uint4 v = lxor(va, vb);
                        if ((dv.w >> kOpShift) == GC_XNOR) v = lxor(v, rl[xinst]);
                        wl[((dv.y & 0xffffu) << ti_log2) + xinst] = v;
                        if (STORE_ALL || (dv.y & kFStoreGlobal))
                            Wt[((size_t)a.gslot[ch.first_desc + p + gl] << ti_log2) + xinst] = v;
                    }
                    p += n;
                    dv = dn;
                }
            } else if (wave < NW) {
                for (uint32_t sx = sidx; sx < ch.nsteps; sx++) {
                    const Step st = read_step(stage_s, sx);
                    const uint32_t rel = st.first - ch.first_desc;
                    for (uint32_t g0 = 0; g0 < st.count; g0 += 64u >> oi_log2) {
                        const uint32_t g = g0 + (lane >> oi_log2);
                        if (g < st.count) {
                            const FDesc d = stage_desc(stage_d, a.descs, direct, rel, st.first, g);
                            uint4 v = lxor(wl[((d.lin & 0xffffu) << ti_log2) + xinst],
                                           wl[((d.lin >> 16) << ti_log2) + xinst]);
                            if ((d.row_op >> kOpShift) == GC_XNOR) v = lxor(v, rl[xinst]);
                            wl[((d.lout & 0xffffu) << ti_log2) + xinst] = v;
                            if (STORE_ALL || (d.lout & kFStoreGlobal))
                                Wt[((size_t)a.gslot[st.first + g] << ti_log2) + xinst] = v;
                        }
                    }
                }
            }
            GC_LPROF(4)
        }
        lds_barrier();  // ends the chunk: XOR results and the freshly staged descriptors become visible
        GC_LPROF(5)
        ch = nx;
    }
    GC_PROF_EPILOGUE()
}

// ------------------------------------------------------------------------------------------------------
template <int NR, int THREADS, int ILP, bool STORE_ALL, bool PROF>
__global__ __launch_bounds__(THREADS) void k_eval_lds(LdsArgs a) {
    constexpr int PF = ((int)kChunkDescs + THREADS - 1) / THREADS;
    GC_LDS_PROLOGUE(false)
    const uint4 *Tt = a.T + (size_t)blockIdx.x * a.t_tile;

    for (uint32_t c = 0; c < a.nchunks; c++) {
        GC_CHUNK_PREFETCH()
        uint32_t sidx = 0, xor_rel0 = 0;
        {
            const Step st = read_step(stage_s, 0);
            const uint32_t rel = st.first - ch.first_desc;
            GC_LPROF(0)
            if (st.nonfree != 0) {
                sidx = 1;
                xor_rel0 = st.count;
                const uint32_t e_and = (st.n_and << ti_log2) << 1;
                const uint32_t e_or = e_and + (st.n_or << ti_log2);
                const uint32_t e_all = e_or + (st.n_inv << ti_log2);
                for (uint32_t t0 = 0; t0 < e_all; t0 += THREADS * ILP) {
                    HPos hp[ILP];
                    FDesc d[ILP];
                    uint4 x[ILP], tab[ILP];
                    uint32_t k[ILP][4];
#pragma unroll
                    for (int j = 0; j < ILP; j++) {
                        hp[j] = hclassify<1, 0, 0>(t0 + j * THREADS + threadIdx.x, e_and, e_or, e_all, st.n_and,
                                                   st.n_or, ti_log2, tim);
                        k[j][0] = k[j][1] = k[j][2] = k[j][3] = 0;
                        x[j] = tab[j] = make_uint4(0, 0, 0, 0);
                        d[j] = FDesc{0, 0, 0, 0};
                        if (hp[j].kind) {
                            d[j] = stage_desc(stage_d, a.descs, direct, rel, st.first, hp[j].g);
                            const uint32_t inst = hp[j].inst, q = hp[j].q;
                            const uint4 *row = Tt + ((size_t)(d[j].row_op & kRowMask) << ti_log2) + inst;
                            const uint4 va = wl[((d[j].lin & 0xffffu) << ti_log2) + inst];
                            x[j] = va;
                            if (hp[j].kind == 1) {
                                if (q) x[j] = wl[((d[j].lin >> 16) << ti_log2) + inst];
                                tab[j] = row[q ? TI : 0];  // issued before the hash: arrives while the AES runs
                                make_k_half(x[j], d[j].tweak + q, k[j]);
                            } else if (hp[j].kind == 3) {
                                tab[j] = row[0];
                                make_k_half(va, d[j].tweak, k[j]);
                            } else {
                                const uint4 vb = wl[((d[j].lin >> 16) << ti_log2) + inst];
                                const uint32_t index = (lbit_s(va) ? 2u : 0u) | (lbit_s(vb) ? 1u : 0u);
                                if (index > 0) tab[j] = row[(size_t)(index - 1) << ti_log2];
                                make_k(va, vb, d[j].tweak, k[j]);
                            }
                        }
                    }
                    uint4 h[ILP];
                    if constexpr (ILP == 2) {
                        if (__any(hp[1].kind != 0)) {
                            hash_dual_n<NR, 2>(k, h, rkr, te, lo);
                        } else {  // this wave only has first-lane work: single AES chain
                            h[0] = hash_dual<NR>(k[0], rkr, te, lo);
                            h[1] = make_uint4(0, 0, 0, 0);
                        }
                    } else {
                        hash_dual_n<NR, ILP>(k, h, rkr, te, lo);
                    }
#pragma unroll
                    for (int j = 0; j < ILP; j++) {
                        const uint32_t kind = hp[j].kind, inst = hp[j].inst, q = hp[j].q;
                        if (kind == 0) continue;
                        uint4 out_label;
                        bool writer = true;
                        if (kind == 1) {  // eval.go:53-78
                            const uint4 av = ldpp128<DPP_PAIR0>(x[j]);
                            uint4 v;
                            if (q) v = lxor(h[j], land(lxor(tab[j], av), smask(x[j])));  // WE = H(b)^(sb ? TE^a : 0)
                            else v = lxor(h[j], land(tab[j], smask(x[j])));              // WG = H(a)^(sa ? TG : 0)
                            out_label = lxor(v, ldpp128<DPP_XOR1>(v));
                            writer = q == 0;
                        } else if (kind == 3) {  // eval.go:96-109
                            out_label = lxor(h[j], land(tab[j], smask(x[j])));
                        } else {  // eval.go:80-94 (tab is zero for index 0)
                            out_label = lxor(h[j], tab[j]);
                        }
                        if (writer) {
                            wl[((d[j].lout & 0xffffu) << ti_log2) + inst] = out_label;
                            if (STORE_ALL || (d[j].lout & kFStoreGlobal))
                                Wt[((size_t)a.gslot[st.first + hp[j].g] << ti_log2) + inst] = out_label;
                        }
                    }
                }
                GC_LPROF(1)
                if (sidx < ch.nsteps) {
                    lds_barrier();
                    GC_LPROF(2)
                }
            }
        }
        // ---- XOR sub-levels (eval.go:49-51), wave-local like in the garbler ----
        GC_CHUNK_COMMIT()
        GC_LPROF(3)
        if (sidx < ch.nsteps) {
            if (wave < NW && !direct) {
                // flat stream over all XOR descriptors of the chunk: an iteration takes up to 64>>oi gates
                // and stops at the next sub-level start (kFLevelStart); the next descriptors are fetched
                // from the LDS stage before the current labels are touched
                const uint32_t gl = lane >> oi_log2;
                uint32_t p = xor_rel0;
                const uint32_t end = ch.ndesc;
                uint4 dv = p + gl < end ? stage_d[p + gl] : make_uint4(0, 0, 0, 0);
                while (p < end) {
                    // lanes whose descriptor belongs to the sub-level of the first lane take part; the label
                    // reads are issued first, the bookkeeping for the next iteration overlaps their latency
                    const uint32_t lvl = __builtin_amdgcn_readfirstlane(dv.z);
                    const bool act = p + gl < end && dv.z == lvl;
                    uint4 va = make_uint4(0, 0, 0, 0), vb = va;
                    if (act) {
                        va = wl[((dv.x & 0xffffu) << ti_log2) + xinst];
                        vb = wl[((dv.x >> 16) << ti_log2) + xinst];
                    }
                    const uint32_t n = (uint32_t)__builtin_popcountll(__ballot(act)) >> oi_log2;
                    const uint4 dn = p + n + gl < end ? stage_d[p + n + gl] : make_uint4(0, 0, 0, 0);
                    if (act) {
                        uint4 v = lxor(va, vb);
                        wl[((dv.y & 0xffffu) << ti_log2) + xinst] = v;
                        if (STORE_ALL || (dv.y & kFStoreGlobal))
                            Wt[((size_t)a.gslot[ch.first_desc + p + gl] << ti_log2) + xinst] = v;
                    }
                    p += n;
                    dv = dn;
                }
            } else if (wave < NW) {
                for (uint32_t sx = sidx; sx < ch.nsteps; sx++) {
                    const Step st = read_step(stage_s, sx);
                    const uint32_t rel = st.first - ch.first_desc;
                    for (uint32_t g0 = 0; g0 < st.count; g0 += 64u >> oi_log2) {
                        const uint32_t g = g0 + (lane >> oi_log2);
                        if (g < st.count) {
                            const FDesc d = stage_desc(stage_d, a.descs, direct, rel, st.first, g);
                            uint4 v = lxor(wl[((d.lin & 0xffffu) << ti_log2) + xinst],
                                           wl[((d.lin >> 16) << ti_log2) + xinst]);
                            wl[((d.lout & 0xffffu) << ti_log2) + xinst] = v;
                            if (STORE_ALL || (d.lout & kFStoreGlobal))
                                Wt[((size_t)a.gslot[st.first + g] << ti_log2) + xinst] = v;
                        }
                    }
                }
            }
            GC_LPROF(4)
        }
        lds_barrier();  // ends the chunk: XOR results and the freshly staged descriptors become visible
        GC_LPROF(5)
        ch = nx;
    }
    GC_PROF_EPILOGUE()
}

size_t fused_lds_bytes(uint32_t nls, uint32_t ti_log2) {
    return (size_t)kStageEnd * sizeof(uint4) + ((size_t)(nls + 1) << ti_log2) * sizeof(uint4);
}

template <typename K>
static hipError_t launch_lds(K kern, int threads, const LdsArgs &a, uint32_t ntiles, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(threads), lds, s, a);
    return hipGetLastError();
}

// Workgroup shape: 1024 threads, ONE AES chain per hash lane.  Measured alternatives on MI355X (aes_128 x
// 1024, AES-256 key, garble pass): 1024 x ILP1 1.12 ms | 512 x ILP2 1.49 ms | 256 x ILP4 2.35 ms | 1024 x ILP2
// with per-wave narrowing 1.26 ms | two 1024-thread workgroups per CU on a 32 KiB table 1.45 ms.  A CDNA4
// SIMD needs ~4 resident waves to keep issuing; ILP inside a wave does not replace them, and a second
// co-resident workgroup slows the serial XOR runs more than its overlap wins back.
hipError_t launch_fused_lds(bool eval, const FusedLdsArgs &f, const BatchGeom &g, hipStream_t s) {
    LdsArgs a{};
    a.descs = f.descs;
    a.gslot = f.gslot;
    a.steps = f.steps;
    a.chunks = f.chunks;
    a.nchunks = f.nchunks;
    a.in_lds = f.in_lds;
    a.nsteps = f.nsteps;
    a.ninputs = f.ninputs;
    a.nls = f.nls;
    a.ti_log2 = g.ti_log2;
    a.w_tile = g.lw.tile_stride;
    a.t_tile = g.lt.tile_stride;
    a.W = f.W;
    a.R = f.R;
    a.T = f.T;
    a.rk = f.rk;
    a.te0 = f.te0;
    a.prof = f.prof;
    if (a.nsteps == 0) return hipSuccess;
    const size_t lds = fused_lds_bytes(f.nls, g.ti_log2);
#define GC_L3(KERN, NR)                                                                      \
    (f.prof ? launch_lds(KERN<NR, 1024, 1, false, true>, 1024, a, g.ntiles, lds, s)              \
            : f.store_all ? launch_lds(KERN<NR, 1024, 1, true, false>, 1024, a, g.ntiles, lds, s) \
                          : launch_lds(KERN<NR, 1024, 1, false, false>, 1024, a, g.ntiles, lds, s))
#define GC_L2(KERN) (f.rounds == 10 ? GC_L3(KERN, 10) : f.rounds == 12 ? GC_L3(KERN, 12) : GC_L3(KERN, 14))
    return eval ? GC_L2(k_eval_lds) : GC_L2(k_garble_lds);
#undef GC_L2
#undef GC_L3
}

}  // namespace gc
