// fused_lds2_kernels.hip — fused LDS-resident schedule with STAGGERED HALF-TILES (production path for TI >= 2).
//
// Same building blocks as fused_lds_kernels.hip (one launch per pass, a 1024-thread workgroup owns a tile of
// TI instances, wire labels live in recycled LDS slots, perm-addressed conflict-free AES table, 2-4 lanes per
// gate-instance joined by DPP, wave-local barrier-free XOR sub-levels), but the tile is split in two halves
// that run ONE STAGE APART:
//
//     interval t:   half 0 executes stage t,   half 1 executes stage t-1
//     stage 2c   = hash stage of chunk c (all table-producing gates of the chunk, LDS-bandwidth bound)
//     stage 2c+1 = XOR stage of chunk c  (its XOR sub-levels: a serial, latency-bound chain per instance)
//
// so in every interval one half hashes on 14-15 waves while the other half's XOR chain runs on 1-2 waves.
// Measured on aes_128 x 1024 the single-phase kernel spent ~22 % of a pass in XOR runs with the LDS pipe
// idle and ~30 % waiting for the ragged second hash pass; here the hash stage of a half fits one pass and
// the XOR run is hidden behind the other half's hashing.  One workgroup barrier per interval.
//
// Staging: hash descriptors of chunk c are needed in intervals 2c, 2c+1, XOR descriptors in 2c+1, 2c+2.
// Chunk c+1 is fetched into registers at the start of interval 2c and committed to the other LDS buffer at
// the start of interval 2c+1 (its previous readers finished in intervals 2c-1 / 2c).
// LDS map: 64 KiB AES table | hash stage 2 x kSHash x 16 B | XOR stage 2 x kSXor x 8 B | R[TI] | wires [slot][TI].
#include "aes_device.h"
#include "kernels.h"

namespace gc {

namespace {

constexpr int T2 = 1024;

constexpr int DPP_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;   // [2,3,0,1]
constexpr int DPP_XOR3 = 0x1B;   // [3,2,1,0]
constexpr int DPP_BC0 = 0x00;    // [0,0,0,0]
constexpr int DPP_BC2 = 0xAA;    // [2,2,2,2]
constexpr int DPP_PAIR0 = 0xA0;  // [0,0,2,2]

template <int CTRL>
__device__ __forceinline__ uint32_t dpp32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ uint4 dpp128(uint4 v) {
    return make_uint4(dpp32<CTRL>(v.x), dpp32<CTRL>(v.y), dpp32<CTRL>(v.z), dpp32<CTRL>(v.w));
}

__device__ __forceinline__ void lds_barrier2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS map in uint4 units
constexpr uint32_t kHOff = kTeDualBytes / 16;
constexpr uint32_t kXOff = kHOff + 2 * kSHash;            // 2 hash buffers of kSHash descriptors
constexpr uint32_t kEnd2 = kXOff + 2 * (kSXor / 2);       // 2 XOR buffers of kSXor 8-byte descriptors

struct L2Args {
    const FDesc *hdescs;
    const uint32_t *hgslot;
    const XDesc *xdescs;
    const uint32_t *xgslot;
    const SChunk *chunks;
    const uint16_t *in_lds;
    uint32_t nchunks, ninputs, ti_log2;
    size_t w_tile, t_tile;
    uint4 *W;
    const uint4 *R;
    uint4 *T;
    const uint32_t *rk;
    const uint32_t *te0;
};

__device__ __forceinline__ SChunk load_chunk(const SChunk *chunks, uint32_t c, uint32_t n) {
    SChunk s{0, 0, 0, 0, 0, 0, {0, 0}};
    if (c < n) {
        s.hfirst = __builtin_amdgcn_readfirstlane(chunks[c].hfirst);
        s.n_and = __builtin_amdgcn_readfirstlane(chunks[c].n_and);
        s.n_or = __builtin_amdgcn_readfirstlane(chunks[c].n_or);
        s.n_inv = __builtin_amdgcn_readfirstlane(chunks[c].n_inv);
        s.xfirst = __builtin_amdgcn_readfirstlane(chunks[c].xfirst);
        s.nx = __builtin_amdgcn_readfirstlane(chunks[c].nx);
    }
    return s;
}

// hash-stage lane -> (kind, gate, instance-in-half, sub-lane); kinds: 0 none, 1 AND, 2 OR, 3 INV
struct HP {
    uint32_t kind, g, inst, q;
};
template <int LQA, int LQO, int LQI>
__device__ __forceinline__ HP hpos(uint32_t t, const SChunk &c, uint32_t th_log2, uint32_t thm) {
    HP p{0, 0, 0, 0};
    const uint32_t e_and = (c.n_and << th_log2) << LQA;
    const uint32_t e_or = e_and + ((c.n_or << th_log2) << LQO);
    const uint32_t e_all = e_or + ((c.n_inv << th_log2) << LQI);
    if (t < e_and) {
        p.kind = 1;
        p.g = t >> (th_log2 + LQA);
        p.inst = (t >> LQA) & thm;
        p.q = t & ((1u << LQA) - 1);
    } else if (t < e_or) {
        const uint32_t u = t - e_and;
        p.kind = 2;
        p.g = c.n_and + (u >> (th_log2 + LQO));
        p.inst = (u >> LQO) & thm;
        p.q = u & ((1u << LQO) - 1);
    } else if (t < e_all) {
        const uint32_t u = t - e_or;
        p.kind = 3;
        p.g = c.n_and + c.n_or + (u >> (th_log2 + LQI));
        p.inst = (u >> LQI) & thm;
        p.q = u & ((1u << LQI) - 1);
    }
    return p;
}
template <int LQA, int LQO, int LQI>
__device__ __forceinline__ uint32_t hlanes(const SChunk &c, uint32_t th_log2) {
    return ((c.n_and << th_log2) << LQA) + ((c.n_or << th_log2) << LQO) + ((c.n_inv << th_log2) << LQI);
}

// XOR stage of one chunk for ONE instance, executed by one wave (lanes = gates of a sub-level); barrier-free:
// a wave's DS operations execute in order (garble.go:331-351 / eval.go:49-51)
template <bool GARBLE, bool STORE_ALL>
__device__ __forceinline__ void xor_stage(const uint2 *xb, uint32_t nx, uint32_t xfirst, const uint32_t *xgslot,
                                          uint4 *wl, const uint4 *rl, uint4 *Wt, uint32_t ti_log2, uint32_t inst,
                                          uint32_t lane) {
    uint32_t p = 0;
    uint2 dv = lane < nx ? xb[lane] : make_uint2(0, 0);
    while (p < nx) {
        const uint32_t lvl = __builtin_amdgcn_readfirstlane(dv.y >> 16);
        const bool act = p + lane < nx && (dv.y >> 16) == lvl;
        uint4 va = make_uint4(0, 0, 0, 0), vb = va;
        if (act) {
            va = wl[((dv.x & 0xffffu) << ti_log2) + inst];
            vb = wl[((dv.x >> 16) << ti_log2) + inst];
        }
        const uint32_t n = (uint32_t)__builtin_popcountll(__ballot(act));
        const uint2 dn = p + n + lane < nx ? xb[p + n + lane] : make_uint2(0, 0);
        if (act) {
            uint4 v = lxor(va, vb);
            if (GARBLE && (dv.y & kXXnor)) v = lxor(v, rl[inst]);
            wl[((dv.y & 0x1fffu) << ti_log2) + inst] = v;
            if (STORE_ALL || (dv.y & kXStoreGlobal)) Wt[((size_t)xgslot[xfirst + p + lane] << ti_log2) + inst] = v;
        }
        p += n;
        dv = dn;
    }
}

#define GC_L2_PROLOGUE(LOAD_R)                                                                               \
    extern __shared__ uint4 smem[];                                                                          \
    uint32_t *te = (uint32_t *)smem;                                                                         \
    const uint32_t ti_log2 = a.ti_log2, TI = 1u << ti_log2, tim = TI - 1;                                    \
    const uint32_t th_log2 = ti_log2 - 1, TH = 1u << th_log2, thm = TH - 1; /* instances per half */          \
    uint4 *hbuf = smem + kHOff;                                                                              \
    uint2 *xbuf = (uint2 *)(smem + kXOff);                                                                   \
    uint4 *rl = smem + kEnd2;                                                                                \
    uint4 *wl = rl + TI;                                                                                     \
    load_te_dual(te, a.te0);                                                                                 \
    uint32_t rkr[4 * (NR + 1)];                                                                              \
    load_round_keys<NR>(rkr, a.rk);                                                                          \
    uint4 *Wt = a.W + (size_t)blockIdx.x * a.w_tile;                                                         \
    if (LOAD_R && threadIdx.x < TI) rl[threadIdx.x] = a.R[(size_t)blockIdx.x * TI + threadIdx.x];           \
    for (uint32_t i = threadIdx.x; i < (a.ninputs << ti_log2); i += T2) {                                    \
        const uint32_t w = i >> ti_log2, ls = a.in_lds[w];                                                   \
        if (ls != 0xffffu) wl[(ls << ti_log2) + (i & tim)] = Wt[i];                                          \
    }                                                                                                        \
    {                                                                                                        \
        const SChunk c0 = load_chunk(a.chunks, 0, a.nchunks);                                                \
        const uint32_t nh0 = c0.n_and + c0.n_or + c0.n_inv;                                                  \
        if (threadIdx.x < nh0) hbuf[threadIdx.x] = ((const uint4 *)a.hdescs)[c0.hfirst + threadIdx.x];       \
        if (threadIdx.x < c0.nx) xbuf[threadIdx.x] = ((const uint2 *)a.xdescs)[c0.xfirst + threadIdx.x];     \
    }                                                                                                        \
    __syncthreads();                                                                                         \
    const uint32_t lo = te_lane_off();                                                                       \
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;         \
    /* XOR workers: one wave per instance of the half when the half has <= 8 instances, else 8 waves */     \
    const uint32_t nwx_log2 = th_log2 < 3 ? th_log2 : 3, NWX = 1u << nwx_log2;                               \
    const uint32_t hash_threads = T2 - NWX * 64, htid = threadIdx.x - NWX * 64;                              \
    uint4 pre_h = make_uint4(0, 0, 0, 0);                                                                    \
    uint2 pre_x = make_uint2(0, 0);                                                                          \
    SChunk pre_c{0, 0, 0, 0, 0, 0, {0, 0}};

// per-interval staging: fetch chunk c+1 in even intervals, commit it in odd ones
#define GC_L2_STAGE()                                                                                        \
    if ((t & 1u) == 0) {                                                                                     \
        pre_c = load_chunk(a.chunks, (t >> 1) + 1, a.nchunks);                                               \
        const uint32_t nh1 = pre_c.n_and + pre_c.n_or + pre_c.n_inv;                                         \
        if (threadIdx.x < nh1) pre_h = ((const uint4 *)a.hdescs)[pre_c.hfirst + threadIdx.x];                \
        if (threadIdx.x < pre_c.nx) pre_x = ((const uint2 *)a.xdescs)[pre_c.xfirst + threadIdx.x];           \
    } else {                                                                                                 \
        const uint32_t nb = ((t >> 1) + 1) & 1u, nh1 = pre_c.n_and + pre_c.n_or + pre_c.n_inv;               \
        if (threadIdx.x < nh1) hbuf[nb * kSHash + threadIdx.x] = pre_h;                                      \
        if (threadIdx.x < pre_c.nx) xbuf[nb * kSXor + threadIdx.x] = pre_x;                                  \
    }

}  // namespace

// ------------------------------------------------------------------------------------------------------------
template <int NR, bool STORE_ALL>
__global__ __launch_bounds__(T2) void k_garble_lds2(L2Args a) {
    GC_L2_PROLOGUE(true)
    uint4 *Tt = a.T + (size_t)blockIdx.x * a.t_tile;
    const uint32_t nint = 2 * a.nchunks + 1;
    for (uint32_t t = 0; t < nint; t++) {
        GC_L2_STAGE()
        const uint32_t hh = t & 1u, xh = hh ^ 1u;          // hashing half / XOR half of this interval
        const uint32_t ch_h = t >> 1;                        // chunk of the hashing half
        const bool xvalid = t >= 1 && ((t - 1) >> 1) < a.nchunks;
        const uint32_t ch_x = xvalid ? (t - 1) >> 1 : 0;
        if (wave < NWX) {
            if (xvalid) {
                const SChunk cx = load_chunk(a.chunks, ch_x, a.nchunks);
                __builtin_amdgcn_s_setprio(3);  // a serial chain: never let it starve behind the hash waves
                for (uint32_t ii = wave; ii < TH; ii += NWX)
                    xor_stage<true, STORE_ALL>(xbuf + (ch_x & 1u) * kSXor, cx.nx, cx.xfirst, a.xgslot, wl, rl, Wt,
                                               ti_log2, xh * TH + ii, lane);
                __builtin_amdgcn_s_setprio(0);
            }
        } else if (ch_h < a.nchunks) {
            const SChunk c = load_chunk(a.chunks, ch_h, a.nchunks);
            const uint4 *hb = hbuf + (ch_h & 1u) * kSHash;
            const uint32_t e_all = hlanes<2, 2, 1>(c, th_log2);
            for (uint32_t t0 = 0; t0 < e_all; t0 += hash_threads) {
                const HP hp = hpos<2, 2, 1>(t0 + htid, c, th_log2, thm);
                if (hp.kind == 0) continue;
                const uint4 dv = hb[hp.g];
                const FDesc d{dv.x, dv.y, dv.z, dv.w};
                const uint32_t inst = hh * TH + hp.inst, q = hp.q;
                const uint4 R = rl[inst];
                const uint4 va = wl[((d.lin & 0xffffu) << ti_log2) + inst];
                uint4 base;
                uint32_t k[4];
                if (hp.kind == 2) {  // OR: e[2u+v] = enc(a_u, b_v, 0, id)  (garble.go:74-83, 421-424)
                    const uint4 vb = wl[((d.lin >> 16) << ti_log2) + inst];
                    const uint4 x = lxor(va, land(R, (q & 2) ? ~0u : 0u));
                    const uint4 y = lxor(vb, land(R, (q & 1) ? ~0u : 0u));
                    base = make_uint4(x.y, y.y, 0, 0);
                    make_k(x, y, d.tweak, k);
                } else {  // AND q=0..3 -> a0,a1,b0,b1 ; INV q=0,1 -> a0,a1 ; K = 2x ^ tweak
                    const bool second = (hp.kind == 1) && (q & 2);
                    base = second ? wl[((d.lin >> 16) << ti_log2) + inst] : va;
                    const uint4 x = lxor(base, land(R, (q & 1) ? ~0u : 0u));
                    make_k_half(x, d.tweak + (second ? 1u : 0u), k);
                }
                const uint4 h = hash_dual<NR>(k, rkr, te, lo);
                uint4 *row = Tt + ((size_t)(d.row_op & kRowMask) << ti_log2) + inst;
                uint4 out_label;
                if (hp.kind == 1) {  // garble.go:353-395
                    const uint4 p = lxor(h, dpp128<DPP_XOR1>(h));  // lanes 0,1: Ha0^Ha1 ; lanes 2,3: Hb0^Hb1
                    const uint4 a0 = dpp128<DPP_BC0>(base);
                    const uint32_t pa = smask(a0);
                    const uint32_t pb = (uint32_t)((int32_t)dpp32<DPP_BC2>(base.y) >> 31);
                    uint4 v, tab;
                    if (q & 2) {
                        tab = lxor(p, a0);                     // TE = Hb0^Hb1^a0
                        v = lxor(h, land(lxor(tab, a0), pb));  // WE0 = Hb0 ^ (pb ? TE^a0 : 0)
                    } else {
                        tab = lxor(p, land(R, pb));            // TG = Ha0^Ha1^(pb?R:0)
                        v = lxor(h, land(tab, pa));            // WG0 = Ha0 ^ (pa ? TG : 0)
                    }
                    out_label = lxor(v, dpp128<DPP_XOR2>(v));
                    if (q == 0) row[0] = tab;
                    else if (q == 2) row[TI] = tab;
                } else if (hp.kind == 3) {  // garble.go:446-474
                    const uint4 p = lxor(h, dpp128<DPP_XOR1>(h));       // E0 ^ E1
                    out_label = lbit_s(base) ? lxor(p, h) : lxor(h, R);  // S(a0) ? E1 : E0^R
                    if (q == 0) row[0] = lxor(p, R);
                } else {  // OR: garble.go:412-444
                    const uint32_t pa = (base.x >> 31) ^ ((q >> 1) & 1), pb = (base.y >> 31) ^ (q & 1);
                    const uint32_t l0 = 2 * pa + pb;
                    const uint4 x1 = dpp128<DPP_XOR1>(h), x2 = dpp128<DPP_XOR2>(h), x3 = dpp128<DPP_XOR3>(h);
                    const uint4 tk = l0 == 0 ? h : l0 == 1 ? x1 : l0 == 2 ? x2 : x3;  // table[q] = e[q ^ l0]
                    const uint4 t0v = dpp128<DPP_BC0>(tk);
                    const uint32_t m0 = l0 == 0 ? ~0u : 0u;
                    const uint4 c0 = lxor(t0v, land(R, ~m0)), c1 = lxor(t0v, land(R, m0));
                    out_label = c0;
                    if (q != 0) row[(size_t)(q - 1) << ti_log2] = lxor(tk, q == l0 ? c0 : c1);
                }
                if (q == 0) {
                    wl[((d.lout & 0xffffu) << ti_log2) + inst] = out_label;
                    if (STORE_ALL || (d.lout & kFStoreGlobal))
                        Wt[((size_t)a.hgslot[c.hfirst + hp.g] << ti_log2) + inst] = out_label;
                }
            }
        }
        lds_barrier2();
    }
}

// ------------------------------------------------------------------------------------------------------------
template <int NR, bool STORE_ALL>
__global__ __launch_bounds__(T2) void k_eval_lds2(L2Args a) {
    GC_L2_PROLOGUE(false)
    const uint4 *Tt = a.T + (size_t)blockIdx.x * a.t_tile;
    const uint32_t nint = 2 * a.nchunks + 1;
    for (uint32_t t = 0; t < nint; t++) {
        GC_L2_STAGE()
        const uint32_t hh = t & 1u, xh = hh ^ 1u;
        const uint32_t ch_h = t >> 1;
        const bool xvalid = t >= 1 && ((t - 1) >> 1) < a.nchunks;
        const uint32_t ch_x = xvalid ? (t - 1) >> 1 : 0;
        if (wave < NWX) {
            if (xvalid) {
                const SChunk cx = load_chunk(a.chunks, ch_x, a.nchunks);
                __builtin_amdgcn_s_setprio(3);
                for (uint32_t ii = wave; ii < TH; ii += NWX)
                    xor_stage<false, STORE_ALL>(xbuf + (ch_x & 1u) * kSXor, cx.nx, cx.xfirst, a.xgslot, wl, rl, Wt,
                                                ti_log2, xh * TH + ii, lane);
                __builtin_amdgcn_s_setprio(0);
            }
        } else if (ch_h < a.nchunks) {
            const SChunk c = load_chunk(a.chunks, ch_h, a.nchunks);
            const uint4 *hb = hbuf + (ch_h & 1u) * kSHash;
            const uint32_t e_all = hlanes<1, 0, 0>(c, th_log2);
            for (uint32_t t0 = 0; t0 < e_all; t0 += hash_threads) {
                const HP hp = hpos<1, 0, 0>(t0 + htid, c, th_log2, thm);
                if (hp.kind == 0) continue;
                const uint4 dv = hb[hp.g];
                const FDesc d{dv.x, dv.y, dv.z, dv.w};
                const uint32_t inst = hh * TH + hp.inst, q = hp.q;
                const uint4 *row = Tt + ((size_t)(d.row_op & kRowMask) << ti_log2) + inst;
                const uint4 va = wl[((d.lin & 0xffffu) << ti_log2) + inst];
                uint4 x = va, tab = make_uint4(0, 0, 0, 0);
                uint32_t k[4];
                if (hp.kind == 1) {
                    if (q) x = wl[((d.lin >> 16) << ti_log2) + inst];
                    tab = row[q ? TI : 0];  // issued before the hash: arrives while the AES runs
                    make_k_half(x, d.tweak + q, k);
                } else if (hp.kind == 3) {
                    tab = row[0];
                    make_k_half(va, d.tweak, k);
                } else {
                    const uint4 vb = wl[((d.lin >> 16) << ti_log2) + inst];
                    const uint32_t index = (lbit_s(va) ? 2u : 0u) | (lbit_s(vb) ? 1u : 0u);
                    if (index > 0) tab = row[(size_t)(index - 1) << ti_log2];
                    make_k(va, vb, d.tweak, k);
                }
                const uint4 h = hash_dual<NR>(k, rkr, te, lo);
                uint4 out_label;
                bool writer = true;
                if (hp.kind == 1) {  // eval.go:53-78
                    const uint4 av = dpp128<DPP_PAIR0>(x);
                    uint4 v;
                    if (q) v = lxor(h, land(lxor(tab, av), smask(x)));  // WE = H(b) ^ (sb ? TE^a : 0)
                    else v = lxor(h, land(tab, smask(x)));              // WG = H(a) ^ (sa ? TG : 0)
                    out_label = lxor(v, dpp128<DPP_XOR1>(v));
                    writer = q == 0;
                } else if (hp.kind == 3) {  // eval.go:96-109
                    out_label = lxor(h, land(tab, smask(x)));
                } else {  // eval.go:80-94 (tab is zero for index 0)
                    out_label = lxor(h, tab);
                }
                if (writer) {
                    wl[((d.lout & 0xffffu) << ti_log2) + inst] = out_label;
                    if (STORE_ALL || (d.lout & kFStoreGlobal))
                        Wt[((size_t)a.hgslot[c.hfirst + hp.g] << ti_log2) + inst] = out_label;
                }
            }
        }
        lds_barrier2();
    }
}

size_t fused_lds2_bytes(uint32_t nls, uint32_t ti_log2) {
    return (size_t)kEnd2 * sizeof(uint4) + ((size_t)(nls + 1) << ti_log2) * sizeof(uint4);
}

template <typename K>
static hipError_t launch2(K kern, const L2Args &a, uint32_t ntiles, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(T2), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_fused_lds2(bool eval, const FusedLds2Args &f, const BatchGeom &g, hipStream_t s) {
    L2Args a{};
    a.hdescs = f.hdescs;
    a.hgslot = f.hgslot;
    a.xdescs = f.xdescs;
    a.xgslot = f.xgslot;
    a.chunks = f.chunks;
    a.in_lds = f.in_lds;
    a.nchunks = f.nchunks;
    a.ninputs = f.ninputs;
    a.ti_log2 = g.ti_log2;
    a.w_tile = g.lw.tile_stride;
    a.t_tile = g.lt.tile_stride;
    a.W = f.W;
    a.R = f.R;
    a.T = f.T;
    a.rk = f.rk;
    a.te0 = f.te0;
    if (a.nchunks == 0) return hipSuccess;
    const size_t lds = fused_lds2_bytes(f.nls, g.ti_log2);
#define GC_M3(KERN, NR) \
    (f.store_all ? launch2(KERN<NR, true>, a, g.ntiles, lds, s) : launch2(KERN<NR, false>, a, g.ntiles, lds, s))
#define GC_M2(KERN) (f.rounds == 10 ? GC_M3(KERN, 10) : f.rounds == 12 ? GC_M3(KERN, 12) : GC_M3(KERN, 14))
    return eval ? GC_M2(k_eval_lds2) : GC_M2(k_garble_lds2);
#undef GC_M2
#undef GC_M3
}

}  // namespace gc
