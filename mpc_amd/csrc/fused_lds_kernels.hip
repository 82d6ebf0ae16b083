// fused_lds_kernels.hip — fused schedule with LDS-RESIDENT wire labels (the production path).
//
// Same tile-per-workgroup structure as fused_kernels.hip, but
//   * the live wire labels of the tile never leave the CU: they sit in LDS slots that the host plan
//     recycles after the last reader (aes_128: <= ~1.1k live labels -> 17 KiB per instance), so the
//     only HBM traffic of a pass is: input labels in, garbled tables out (garble) / in (eval), output
//     labels out;
//   * steps follow the hash-phase schedule (plan.h): all table-producing gates of one non-free depth are
//     hashed together (89 phases for aes_128) and the XOR sub-levels in between touch LDS only;
//   * the inter-step barrier waits for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier) — table
//     stores and descriptor prefetches stay in flight across it.
//   * descriptors never stall a step: the plan cuts the schedule into chunks (one hash phase + its XOR
//     sub-levels); while chunk c executes, every thread prefetches one descriptor of chunk c+1 into
//     registers and drops it into an LDS staging area at the chunk boundary.
// LDS map (dynamic): [0, 64 KiB) perm-addressed dual AES table | 16 KiB descriptor stage | 2 KiB step
// stage | R of the tile | wire slots.
#include "aes_device.h"
#include "kernels.h"

namespace gc {

constexpr int kLdsThreads = 1024;

constexpr int DPP_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;   // [2,3,0,1]
constexpr int DPP_XOR3 = 0x1B;   // [3,2,1,0]
constexpr int DPP_BC0 = 0x00;    // [0,0,0,0]
constexpr int DPP_BC2 = 0xAA;    // [2,2,2,2]
constexpr int DPP_PAIR0 = 0xA0;  // [0,0,2,2]

template <int CTRL>
__device__ __forceinline__ uint32_t ldpp32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ uint4 ldpp128(uint4 v) {
    return make_uint4(ldpp32<CTRL>(v.x), ldpp32<CTRL>(v.y), ldpp32<CTRL>(v.z), ldpp32<CTRL>(v.w));
}

struct LPos {
    int kind;  // 0 none, 1 AND, 2 OR, 3 INV, 4 free
    uint32_t g, inst, q;
};

template <int LQA, int LQO, int LQI>
__device__ __forceinline__ LPos lclassify(const Step &st, uint32_t t, uint32_t ti_log2, uint32_t tim) {
    LPos p{0, 0, 0, 0};
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
__device__ __forceinline__ uint32_t llanes(const Step &st, uint32_t ti_log2) {
    return ((st.n_and << ti_log2) << LQA) + ((st.n_or << ti_log2) << LQO) + ((st.n_inv << ti_log2) << LQI) +
           ((st.count - st.nonfree) << ti_log2);
}

// workgroup barrier that only drains LDS traffic (global stores / prefetches keep flying)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define GC_LPROF(slot)                                               \
    if constexpr (PROF) {                                            \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  \
        const uint64_t now__ = __builtin_amdgcn_s_memtime();         \
        pacc[slot] += now__ - plast;                                 \
        plast = now__;                                               \
    }

constexpr uint32_t kStageDescOff = kTeDualBytes / 16;                  // in uint4 units
constexpr uint32_t kStageStepOff = kStageDescOff + kChunkDescs;         // 1024 descriptors x 16 B
constexpr uint32_t kStageEnd = kStageStepOff + kChunkSteps * 2;         // 64 steps x 32 B

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

template <int NR, bool STORE_ALL, bool PROF>
__global__ __launch_bounds__(kLdsThreads) void k_garble_lds(LdsArgs a) {
    extern __shared__ uint4 smem[];
    uint32_t *te = (uint32_t *)smem;
    const uint32_t ti_log2 = a.ti_log2, TI = 1u << ti_log2, tim = TI - 1;
    uint4 *stage_d = smem + kStageDescOff;
    uint4 *stage_s = smem + kStageStepOff;
    uint4 *rl = smem + kStageEnd;  // R of the tile
    uint4 *wl = rl + TI;           // wire slots [slot][TI]
    load_te_dual(te, a.te0);
    uint32_t rkr[4 * (NR + 1)];
    load_round_keys<NR>(rkr, a.rk);
    uint4 *Wt = a.W + (size_t)blockIdx.x * a.w_tile;
    uint4 *Tt = a.T + (size_t)blockIdx.x * a.t_tile;
    if (threadIdx.x < TI) rl[threadIdx.x] = a.R[(size_t)blockIdx.x * TI + threadIdx.x];
    for (uint32_t i = threadIdx.x; i < (a.ninputs << ti_log2); i += kLdsThreads) {
        const uint32_t w = i >> ti_log2, ls = a.in_lds[w];
        if (ls != 0xffffu) wl[(ls << ti_log2) + (i & tim)] = Wt[i];  // global input slots are [w][TI]
    }
    Chunk ch = read_chunk(a.chunks, 0, a.nchunks);
    if (threadIdx.x < ch.ndesc && ch.ndesc <= kChunkDescs)
        stage_d[threadIdx.x] = ((const uint4 *)a.descs)[ch.first_desc + threadIdx.x];
    if (threadIdx.x < ch.nsteps) {
        const Step sv = a.steps[ch.first_step + threadIdx.x];
        stage_s[2 * threadIdx.x] = make_uint4(sv.first, sv.count, sv.nonfree, sv.n_and);
        stage_s[2 * threadIdx.x + 1] = make_uint4(sv.n_or, sv.n_inv, 0, 0);
    }
    __syncthreads();
    const uint32_t lo = te_lane_off();
    uint64_t pacc[4] = {0, 0, 0, 0}, plast = 0;
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();

    for (uint32_t c = 0; c < a.nchunks; c++) {
      // prefetch chunk c+1 (one descriptor + one step record per thread) — lands in LDS at the chunk end
      const Chunk nx = read_chunk(a.chunks, c + 1, a.nchunks);
      const bool stage_next = nx.ndesc <= kChunkDescs;
      uint4 pre_d = make_uint4(0, 0, 0, 0);
      Step pre_s{0, 0, 0, 0, 0, 0};
      if (threadIdx.x < nx.ndesc && stage_next) pre_d = ((const uint4 *)a.descs)[nx.first_desc + threadIdx.x];
      if (threadIdx.x < nx.nsteps) pre_s = a.steps[nx.first_step + threadIdx.x];
      const bool direct = ch.ndesc > kChunkDescs;
      for (uint32_t sidx = 0; sidx < ch.nsteps; sidx++) {
        const Step st = read_step(stage_s, sidx);
        const uint32_t rel = st.first - ch.first_desc;
        const uint32_t e_all = llanes<2, 2, 1>(st, ti_log2);
        for (uint32_t t0 = 0; t0 < e_all; t0 += kLdsThreads) {
            const LPos lp = lclassify<2, 2, 1>(st, t0 + threadIdx.x, ti_log2, tim);
            const int kind = lp.kind;
            const uint32_t g = lp.g, inst = lp.inst, q = lp.q;
            if (kind == 0) continue;
            FDesc d;
            if (direct) d = a.descs[st.first + g];
            else {
                const uint4 dv = stage_d[rel + g];
                d = FDesc{dv.x, dv.y, dv.z, dv.w};
            }
            GC_LPROF(0)
            const uint32_t l0s = d.lin & 0xffffu, l1s = d.lin >> 16, los = d.lout & 0xffffu;
            const bool to_global = STORE_ALL || (d.lout & kFStoreGlobal);
            const uint4 va = wl[(l0s << ti_log2) + inst];
            if (kind == 4) {
                uint4 v = lxor(va, wl[(l1s << ti_log2) + inst]);
                if ((d.row_op >> kOpShift) == GC_XNOR) v = lxor(v, rl[inst]);  // garble.go:342-351
                wl[(los << ti_log2) + inst] = v;
                if (to_global) Wt[((size_t)a.gslot[st.first + g] << ti_log2) + inst] = v;
                continue;
            }
            // ---- hash lanes ----
            const uint4 R = rl[inst];
            uint4 base;
            uint32_t k[4];
            if (kind == 2) {  // OR: e[2u+v] = enc(a_u, b_v, 0, id), K = 2a ^ 4b ^ id  (garble.go:74-83)
                const uint4 vb = wl[(l1s << ti_log2) + inst];
                const uint4 x = lxor(va, land(R, (q & 2) ? ~0u : 0u));
                const uint4 y = lxor(vb, land(R, (q & 1) ? ~0u : 0u));
                base = make_uint4(x.y, y.y, 0, 0);
                make_k(x, y, d.tweak, k);
            } else {  // AND q=0..3 -> a0,a1,b0,b1 ; INV q=0,1 -> a0,a1 ; K = 2x ^ tweak
                const bool second = (kind == 1) && (q & 2);
                base = second ? wl[(l1s << ti_log2) + inst] : va;
                const uint4 x = lxor(base, land(R, (q & 1) ? ~0u : 0u));
                make_k_half(x, d.tweak + (second ? 1u : 0u), k);
            }
            GC_LPROF(1)
            const uint4 h = hash_dual<NR>(k, rkr, te, lo);
            uint4 *row = Tt + ((size_t)(d.row_op & kRowMask) << ti_log2) + inst;
            uint4 out_label;
            bool writer = false;
            if (kind == 1) {  // garble.go:353-395
                const uint4 p = lxor(h, ldpp128<DPP_XOR1>(h));  // lanes 0,1: Ha0^Ha1 ; lanes 2,3: Hb0^Hb1
                const uint4 a0 = ldpp128<DPP_BC0>(base);
                const uint32_t pa = smask(a0);
                const uint32_t pb = (uint32_t)((int32_t)ldpp32<DPP_BC2>(base.y) >> 31);
                uint4 v, tab;
                if (q & 2) {
                    tab = lxor(p, a0);                     // TE = Hb0^Hb1^a0
                    v = lxor(h, land(lxor(tab, a0), pb));  // WE0 = Hb0 ^ (pb ? TE^a0 : 0)
                } else {
                    tab = lxor(p, land(R, pb));            // TG = Ha0^Ha1^(pb?R:0)
                    v = lxor(h, land(tab, pa));            // WG0 = Ha0 ^ (pa ? TG : 0)
                }
                const uint4 other = ldpp128<DPP_XOR2>(v);
                out_label = lxor(v, other);
                writer = q == 0;
                if (q == 0) row[0] = tab;
                else if (q == 2) row[TI] = tab;
            } else if (kind == 3) {  // garble.go:446-474
                const uint4 p = lxor(h, ldpp128<DPP_XOR1>(h));  // E0 ^ E1
                out_label = lbit_s(base) ? lxor(p, h) : lxor(h, R);  // S(a0) ? E1 : E0^R   (lane 0)
                writer = q == 0;
                if (q == 0) row[0] = lxor(p, R);
            } else {  // OR: garble.go:412-444
                const uint32_t pa = (base.x >> 31) ^ ((q >> 1) & 1), pb = (base.y >> 31) ^ (q & 1);
                const uint32_t l0 = 2 * pa + pb;
                const uint4 x1 = ldpp128<DPP_XOR1>(h), x2 = ldpp128<DPP_XOR2>(h), x3 = ldpp128<DPP_XOR3>(h);
                const uint4 tk = l0 == 0 ? h : l0 == 1 ? x1 : l0 == 2 ? x2 : x3;  // table[q] = e[q ^ l0]
                const uint4 t0v = ldpp128<DPP_BC0>(tk);
                const uint32_t m0 = l0 == 0 ? ~0u : 0u;
                const uint4 c0 = lxor(t0v, land(R, ~m0)), c1 = lxor(t0v, land(R, m0));
                out_label = c0;
                writer = q == 0;
                if (q != 0) row[(size_t)(q - 1) << ti_log2] = lxor(tk, q == l0 ? c0 : c1);
            }
            if (writer) {
                wl[(los << ti_log2) + inst] = out_label;
                if (to_global) Wt[((size_t)a.gslot[st.first + g] << ti_log2) + inst] = out_label;
            }
        }
        GC_LPROF(2)
        lds_barrier();
        GC_LPROF(3)
      }
      // chunk boundary: every reader of the staging area is past the barrier above
      if (threadIdx.x < nx.ndesc && stage_next) stage_d[threadIdx.x] = pre_d;
      if (threadIdx.x < nx.nsteps) {
          stage_s[2 * threadIdx.x] = make_uint4(pre_s.first, pre_s.count, pre_s.nonfree, pre_s.n_and);
          stage_s[2 * threadIdx.x + 1] = make_uint4(pre_s.n_or, pre_s.n_inv, 0, 0);
      }
      lds_barrier();
      ch = nx;
    }
    if constexpr (PROF) {
        if (threadIdx.x == 0 || threadIdx.x == kLdsThreads - 64)
            for (int i = 0; i < 4; i++) a.prof[(size_t)blockIdx.x * 8 + (threadIdx.x ? 4 : 0) + i] = pacc[i];
    }
}

template <int NR, bool STORE_ALL, bool PROF>
__global__ __launch_bounds__(kLdsThreads) void k_eval_lds(LdsArgs a) {
    extern __shared__ uint4 smem[];
    uint32_t *te = (uint32_t *)smem;
    const uint32_t ti_log2 = a.ti_log2, TI = 1u << ti_log2, tim = TI - 1;
    uint4 *stage_d = smem + kStageDescOff;
    uint4 *stage_s = smem + kStageStepOff;
    uint4 *wl = smem + kStageEnd + TI;  // same map as the garbler (R slot unused)
    load_te_dual(te, a.te0);
    uint32_t rkr[4 * (NR + 1)];
    load_round_keys<NR>(rkr, a.rk);
    uint4 *Wt = a.W + (size_t)blockIdx.x * a.w_tile;
    const uint4 *Tt = a.T + (size_t)blockIdx.x * a.t_tile;
    for (uint32_t i = threadIdx.x; i < (a.ninputs << ti_log2); i += kLdsThreads) {
        const uint32_t w = i >> ti_log2, ls = a.in_lds[w];
        if (ls != 0xffffu) wl[(ls << ti_log2) + (i & tim)] = Wt[i];
    }
    Chunk ch = read_chunk(a.chunks, 0, a.nchunks);
    if (threadIdx.x < ch.ndesc && ch.ndesc <= kChunkDescs)
        stage_d[threadIdx.x] = ((const uint4 *)a.descs)[ch.first_desc + threadIdx.x];
    if (threadIdx.x < ch.nsteps) {
        const Step sv = a.steps[ch.first_step + threadIdx.x];
        stage_s[2 * threadIdx.x] = make_uint4(sv.first, sv.count, sv.nonfree, sv.n_and);
        stage_s[2 * threadIdx.x + 1] = make_uint4(sv.n_or, sv.n_inv, 0, 0);
    }
    __syncthreads();
    const uint32_t lo = te_lane_off();
    uint64_t pacc[4] = {0, 0, 0, 0}, plast = 0;
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();

    for (uint32_t c = 0; c < a.nchunks; c++) {
      const Chunk nx = read_chunk(a.chunks, c + 1, a.nchunks);
      const bool stage_next = nx.ndesc <= kChunkDescs;
      uint4 pre_d = make_uint4(0, 0, 0, 0);
      Step pre_s{0, 0, 0, 0, 0, 0};
      if (threadIdx.x < nx.ndesc && stage_next) pre_d = ((const uint4 *)a.descs)[nx.first_desc + threadIdx.x];
      if (threadIdx.x < nx.nsteps) pre_s = a.steps[nx.first_step + threadIdx.x];
      const bool direct = ch.ndesc > kChunkDescs;
      for (uint32_t sidx = 0; sidx < ch.nsteps; sidx++) {
        const Step st = read_step(stage_s, sidx);
        const uint32_t rel = st.first - ch.first_desc;
        const uint32_t e_all = llanes<1, 0, 0>(st, ti_log2);
        for (uint32_t t0 = 0; t0 < e_all; t0 += kLdsThreads) {
            const LPos lp = lclassify<1, 0, 0>(st, t0 + threadIdx.x, ti_log2, tim);
            const int kind = lp.kind;
            const uint32_t g = lp.g, inst = lp.inst, q = lp.q;
            if (kind == 0) continue;
            FDesc d;
            if (direct) d = a.descs[st.first + g];
            else {
                const uint4 dv = stage_d[rel + g];
                d = FDesc{dv.x, dv.y, dv.z, dv.w};
            }
            GC_LPROF(0)
            const uint32_t l0s = d.lin & 0xffffu, l1s = d.lin >> 16, los = d.lout & 0xffffu;
            const bool to_global = STORE_ALL || (d.lout & kFStoreGlobal);
            const uint4 va = wl[(l0s << ti_log2) + inst];
            if (kind == 4) {  // eval.go:49-51
                const uint4 v = lxor(va, wl[(l1s << ti_log2) + inst]);
                wl[(los << ti_log2) + inst] = v;
                if (to_global) Wt[((size_t)a.gslot[st.first + g] << ti_log2) + inst] = v;
                continue;
            }
            const uint4 *row = Tt + ((size_t)(d.row_op & kRowMask) << ti_log2) + inst;
            uint32_t k[4];
            uint4 x = va, vb = make_uint4(0, 0, 0, 0), tab = make_uint4(0, 0, 0, 0);
            if (kind == 1) {
                if (q) x = wl[(l1s << ti_log2) + inst];
                tab = row[q ? TI : 0];  // issued before the hash: arrives while the AES runs
                make_k_half(x, d.tweak + q, k);
            } else if (kind == 3) {
                tab = row[0];
                make_k_half(x, d.tweak, k);
            } else {
                vb = wl[(l1s << ti_log2) + inst];
                const uint32_t index = (lbit_s(va) ? 2u : 0u) | (lbit_s(vb) ? 1u : 0u);
                if (index > 0) tab = row[(size_t)(index - 1) << ti_log2];
                make_k(va, vb, d.tweak, k);
            }
            GC_LPROF(1)
            const uint4 h = hash_dual<NR>(k, rkr, te, lo);
            uint4 out_label;
            bool writer = true;
            if (kind == 1) {  // eval.go:53-78
                const uint4 av = ldpp128<DPP_PAIR0>(x);
                uint4 v;
                if (q) v = lxor(h, land(lxor(tab, av), smask(x)));  // WE = H(b) ^ (sb ? TE^a : 0)
                else v = lxor(h, land(tab, smask(x)));              // WG = H(a) ^ (sa ? TG : 0)
                out_label = lxor(v, ldpp128<DPP_XOR1>(v));
                writer = q == 0;
            } else if (kind == 3) {  // eval.go:96-109
                out_label = lxor(h, land(tab, smask(x)));
            } else {  // eval.go:80-94 (tab is zero for index 0)
                out_label = lxor(h, tab);
            }
            if (writer) {
                wl[(los << ti_log2) + inst] = out_label;
                if (to_global) Wt[((size_t)a.gslot[st.first + g] << ti_log2) + inst] = out_label;
            }
        }
        GC_LPROF(2)
        lds_barrier();
        GC_LPROF(3)
      }
      // chunk boundary: every reader of the staging area is past the barrier above
      if (threadIdx.x < nx.ndesc && stage_next) stage_d[threadIdx.x] = pre_d;
      if (threadIdx.x < nx.nsteps) {
          stage_s[2 * threadIdx.x] = make_uint4(pre_s.first, pre_s.count, pre_s.nonfree, pre_s.n_and);
          stage_s[2 * threadIdx.x + 1] = make_uint4(pre_s.n_or, pre_s.n_inv, 0, 0);
      }
      lds_barrier();
      ch = nx;
    }
    if constexpr (PROF) {
        if (threadIdx.x == 0 || threadIdx.x == kLdsThreads - 64)
            for (int i = 0; i < 4; i++) a.prof[(size_t)blockIdx.x * 8 + (threadIdx.x ? 4 : 0) + i] = pacc[i];
    }
}

size_t fused_lds_bytes(uint32_t nls, uint32_t ti_log2) {
    return (size_t)kStageEnd * sizeof(uint4) + ((size_t)(nls + 1) << ti_log2) * sizeof(uint4);
}

template <typename K>
static hipError_t launch_lds(K kern, const LdsArgs &a, uint32_t ntiles, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(kLdsThreads), lds, s, a);
    return hipGetLastError();
}

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
#define GC_L3(KERN, NR)                                                                       \
    (f.prof ? launch_lds(KERN<NR, false, true>, a, g.ntiles, lds, s)                           \
            : f.store_all ? launch_lds(KERN<NR, true, false>, a, g.ntiles, lds, s)            \
                          : launch_lds(KERN<NR, false, false>, a, g.ntiles, lds, s))
#define GC_L2(KERN) (f.rounds == 10 ? GC_L3(KERN, 10) : f.rounds == 12 ? GC_L3(KERN, 12) : GC_L3(KERN, 14))
    return eval ? GC_L2(k_eval_lds) : GC_L2(k_garble_lds);
#undef GC_L2
#undef GC_L3
}

}  // namespace gc
