// gc_kernels.hip — CDNA4 (gfx950) kernels for Circuit.Garble / Circuit.Eval.
//
// Mapping: one thread = one gate of one instance.  Lanes of a wavefront run along the
// INSTANCE axis, so every label access of a wave is 64 consecutive uint4 = one 1 KiB
// coalesced transaction, the gate descriptor is wave-uniform (scalar loads, no divergence)
// and the AES round keys live in SGPRs.  The four T-tables sit in LDS (4 KiB per block).
// Arithmetic restated from circuit/garble.go:311-482 (garbleInto) and circuit/eval.go:28-112.
#include <cstdlib>

#include "aes_device.h"
#include "kernels.h"
#include "level_gate.h"

namespace gc {

BatchGeom make_geom(uint32_t batch, int schedule, uint32_t nslots, uint32_t slab_rows, uint32_t nls,
                    uint32_t max_ti_log2, bool flat, uint32_t ustride) {
    BatchGeom g{};
    g.batch = batch;
    if (batch >= 256) {
        g.lg = 8;
        g.yblocks = (batch + 255) / 256;
    } else {
        uint32_t lg = 0;
        while ((1u << lg) < batch) lg++;
        g.lg = lg;
        g.yblocks = 1;
    }
    if (schedule == 0) {
        g.bstride = (batch + 63u) & ~63u;
        g.ti_log2 = 31;
        g.ntiles = 1;
        g.lw = Layout{31, 0x7fffffffu, g.bstride, 0};
        g.lt = Layout{31, 0x7fffffffu, g.bstride, 0};
    } else {
        // instances per workgroup tile: aim at >= 256 workgroups (one per CU), at most 64 lanes wide
        uint32_t t = 0;
        while (t < max_ti_log2 && (batch >> (t + 1)) >= 256) t++;
        // LDS-resident wires: the tile's live labels (+R) must fit beside the 64 KiB AES table
        const size_t lds_budget = 160 * 1024;
        auto need = [&](uint32_t tt) { return flat ? fused_flat_bytes(nls, tt, ustride) : fused_lds_bytes(nls, tt); };
        g.lds_wires = nls != 0xffffffffu && need(0) <= lds_budget;
        if (g.lds_wires)
            while (t > 0 && need(t) > lds_budget) t--;
        g.ti_log2 = t;
        const uint32_t ti = 1u << t;
        g.ntiles = (batch + ti - 1) / ti;
        g.bstride = g.ntiles * ti;
        g.lw = Layout{t, ti - 1, ti, (size_t)nslots * ti};
        g.lt = Layout{t, ti - 1, ti, (size_t)(slab_rows ? slab_rows : 1) * ti};
    }
    return g;
}

struct ThreadPos {
    uint32_t gate, inst;
};

// gate / instance of this thread; when the instance tile covers whole waves (lg >= 6) the gate
// index is made provably wave-uniform so that the descriptor comes in through scalar loads.
template <bool UNIFORM>
__device__ __forceinline__ ThreadPos thread_pos(uint32_t lg) {
    ThreadPos p;
    uint32_t gl = threadIdx.x >> lg;
    if (UNIFORM) gl = __builtin_amdgcn_readfirstlane(gl);
    p.gate = blockIdx.x * (256u >> lg) + gl;
    p.inst = blockIdx.y * 256u + (threadIdx.x & ((1u << lg) - 1u));
    return p;
}

// ------------------------------------------------------------------------------------------
// Garble one step (a set of mutually independent gates) for all instances.
// ------------------------------------------------------------------------------------------
template <int NR, bool UNIFORM>
__global__ __launch_bounds__(256) void k_garble_level(const GateDesc *__restrict__ descs, uint32_t count,
                                                      uint32_t nonfree, uint32_t out_slot0, uint32_t batch,
                                                      uint32_t bstride, uint32_t lg, uint4 *__restrict__ W,
                                                      const uint4 *__restrict__ Rv, uint4 *__restrict__ T,
                                                      const uint32_t *__restrict__ rk,
                                                      const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeWords];
    const bool need_tables = blockIdx.x * (256u >> lg) < nonfree;  // block-uniform
    if (need_tables) {
        load_te_tables(te, g_te0);
        __syncthreads();
    }
    const ThreadPos tp = thread_pos<UNIFORM>(lg);
    if (tp.gate >= count || tp.inst >= batch) return;
    const GateDesc d = descs[tp.gate];
    garble_one<NR>(d, tp.inst, bstride, W, Rv[tp.inst], T, W + (size_t)(out_slot0 + tp.gate) * bstride + tp.inst, rk, te);
}

// ------------------------------------------------------------------------------------------
// Evaluate one step for all instances (circuit/eval.go:28-112).
// ------------------------------------------------------------------------------------------
template <int NR, bool UNIFORM>
__global__ __launch_bounds__(256) void k_eval_level(const GateDesc *__restrict__ descs, uint32_t count,
                                                    uint32_t nonfree, uint32_t out_slot0, uint32_t batch,
                                                    uint32_t bstride, uint32_t lg, uint4 *__restrict__ W,
                                                    const uint4 *__restrict__ T, const uint32_t *__restrict__ rk,
                                                    const uint32_t *__restrict__ g_te0) {
    __shared__ uint32_t te[kTeWords];
    const bool need_tables = blockIdx.x * (256u >> lg) < nonfree;
    if (need_tables) {
        load_te_tables(te, g_te0);
        __syncthreads();
    }
    const ThreadPos tp = thread_pos<UNIFORM>(lg);
    if (tp.gate >= count || tp.inst >= batch) return;
    const GateDesc d = descs[tp.gate];
    eval_one<NR>(d, tp.inst, bstride, W, T, W + (size_t)(out_slot0 + tp.gate) * bstride + tp.inst, rk, te);
}

// ------------------------------------------------------------------------------------------
// The same level with the hashes of a gate SPREAD OVER WAVES (round 6).  In the kernels above a thread garbles a whole AND:
// four interleaved AES blocks, ~2 800 instructions of ONE wave per level — measured 6 us of an 8 us graph node on aes_128 x
// 1 024 (a dependent EMPTY node costs 1.6 us, one that moves the level's 6.5 MB 2.0 us: profiles/r06_graph_node_ubench.txt),
// i.e. a level was its busiest wave's instruction stream, not a launch.  Here a workgroup owns ONE table-producing gate and 64
// instances (the evaluator: 128) and wave q computes hash q of the gate for them — garbler AND: H(a0), H(a1), H(b0), H(b1);
// INV: two; OR: its four encryptions — the lanes still run along the instance axis (every label access one coalesced 1 KiB
// read), the hashes meet in 3 KiB of LDS and wave 0 does the Half-Gates combine exactly as garble_one / eval_one (level_gate.h)
// do.  The free gates of the level stay thread = (gate, instance), in the workgroups behind the hash workgroups of the SAME
// launch: still one data-parallel launch per dependency level.
// ------------------------------------------------------------------------------------------
template <int NR>
__global__ __launch_bounds__(256) void k_garble_level_split(const GateDesc *__restrict__ descs, uint32_t count, uint32_t nonfree,
                                                            uint32_t out_slot0, uint32_t batch, uint32_t bstride, uint32_t lg,
                                                            uint4 *__restrict__ W, const uint4 *__restrict__ Rv,
                                                            uint4 *__restrict__ T, const uint32_t *__restrict__ rk,
                                                            const uint32_t *__restrict__ g_te0, uint32_t chunks, uint32_t nb_hash,
                                                            uint32_t gx_free) {
    __shared__ uint32_t te[kTeWords];
    __shared__ uint4 xh[3][64];
    if (blockIdx.x >= nb_hash) {  // the level's free gates: thread = (gate, instance)
        const uint32_t b = blockIdx.x - nb_hash, bx = b % gx_free, by = b / gx_free;
        const uint32_t gate = nonfree + bx * (256u >> lg) + (threadIdx.x >> lg);
        const uint32_t inst = by * 256u + (threadIdx.x & ((1u << lg) - 1u));
        if (gate >= count || inst >= batch) return;
        const GateDesc d = descs[gate];
        garble_one<NR>(d, inst, bstride, W, Rv[inst], T, W + (size_t)(out_slot0 + gate) * bstride + inst, rk, te);
        return;
    }
    const uint32_t gate = blockIdx.x / chunks, chunk = blockIdx.x - gate * chunks;
    const uint32_t q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t inst = chunk * 64u + lane;
    const bool live = inst < batch;
    const size_t i = live ? inst : batch - 1;
    // the operands' loads go out first: their round trip to HBM runs under the table load and its barrier
    const GateDesc d = descs[gate];
    const uint32_t op = d.row_op >> kOpShift;
    const uint4 R = Rv[i];
    const uint4 a0 = W[(size_t)d.in0 * bstride + i];
    const uint4 b0 = op == GC_INV ? a0 : W[(size_t)d.in1 * bstride + i];
    load_te_tables(te, g_te0);
    __syncthreads();
    uint4 h[1] = {make_uint4(0, 0, 0, 0)};
    if (op != GC_INV || q < 2) {
        uint32_t k[1][4];
        if (op == GC_OR) {  // e[2u + v] = enc(a_u, b_v, 0, id): wave q = 2u + v
            make_k(lxor(a0, land(R, (q & 2) ? ~0u : 0u)), lxor(b0, land(R, (q & 1) ? ~0u : 0u)), d.tweak, k[0]);
        } else {  // AND: H(a0), H(a1) under the gate's tweak, H(b0), H(b1) under tweak + 1; INV: H(a0), H(a1)
            const uint4 x = lxor((op == GC_AND && (q & 2)) ? b0 : a0, land(R, (q & 1) ? ~0u : 0u));
            make_k_half(x, d.tweak + (op == GC_AND ? (q >> 1) : 0u), k[0]);
        }
        hash_n<NR, 1>(k, h, rk, te);
    }
    if (q > 0) xh[q - 1][lane] = h[0];
    __syncthreads();
    if (q != 0 || !live) return;
    const uint4 h0 = h[0], h1 = xh[0][lane], h2 = xh[1][lane], h3 = xh[2][lane];
    uint4 *out = W + (size_t)(out_slot0 + gate) * bstride + inst;
    uint4 *row = T + (size_t)(d.row_op & kRowMask) * bstride + inst;
    if (op == GC_AND) {  // garble.go:353-395
        const uint32_t pa = smask(a0), pb = smask(b0);
        const uint4 tg = lxor(lxor(h0, h1), land(R, pb));
        const uint4 wg0 = lxor(h0, land(tg, pa));
        const uint4 te_ = lxor(lxor(h2, h3), a0);
        const uint4 we0 = lxor(h2, land(lxor(te_, a0), pb));
        *out = lxor(wg0, we0);
        row[0] = tg;
        row[bstride] = te_;
    } else if (op == GC_INV) {  // garble.go:446-474
        *out = lbit_s(a0) ? h1 : lxor(h0, R);
        row[0] = lxor(lxor(h0, h1), R);
    } else {  // GC_OR: garble.go:412-444 (as garble_one)
        const uint4 e[4] = {h0, h1, h2, h3};
        const uint32_t pa = lbit_s(a0) ? 1u : 0u, pb = lbit_s(b0) ? 1u : 0u;
        const uint32_t l0 = 2u * pa + pb;
        const uint32_t m0 = l0 == 0 ? ~0u : 0u, m1 = l0 == 1 ? ~0u : 0u, m2 = l0 == 2 ? ~0u : 0u, m3 = l0 == 3 ? ~0u : 0u;
        auto pick = [&](uint32_t ma, uint32_t mb, uint32_t mc, uint32_t md) {
            return lxor(lxor(land(e[0], ma), land(e[1], mb)), lxor(land(e[2], mc), land(e[3], md)));
        };
        const uint4 t0 = pick(m0, m1, m2, m3), t1 = pick(m1, m0, m3, m2), t2 = pick(m2, m3, m0, m1), t3 = pick(m3, m2, m1, m0);
        const uint4 c0 = lxor(t0, land(R, ~m0)), c1 = lxor(t0, land(R, m0));
        row[0] = lxor(t1, lxor(land(c0, m1), land(c1, ~m1)));
        row[bstride] = lxor(t2, lxor(land(c0, m2), land(c1, ~m2)));
        row[2 * (size_t)bstride] = lxor(t3, lxor(land(c0, m3), land(c1, ~m3)));
        *out = c0;
    }
}

// the evaluator: an AND has two hashes (eval.go:53-78), INV and OR one: a workgroup owns one gate and 128 instances — waves 0 / 1
// the first 64 (hashes of a and b), waves 2 / 3 the second
template <int NR>
__global__ __launch_bounds__(256) void k_eval_level_split(const GateDesc *__restrict__ descs, uint32_t count, uint32_t nonfree,
                                                          uint32_t out_slot0, uint32_t batch, uint32_t bstride, uint32_t lg,
                                                          uint4 *__restrict__ W, const uint4 *__restrict__ T,
                                                          const uint32_t *__restrict__ rk, const uint32_t *__restrict__ g_te0,
                                                          uint32_t chunks, uint32_t nb_hash, uint32_t gx_free) {
    __shared__ uint32_t te[kTeWords];
    __shared__ uint4 xh[2][64];
    if (blockIdx.x >= nb_hash) {
        const uint32_t b = blockIdx.x - nb_hash, bx = b % gx_free, by = b / gx_free;
        const uint32_t gate = nonfree + bx * (256u >> lg) + (threadIdx.x >> lg);
        const uint32_t inst = by * 256u + (threadIdx.x & ((1u << lg) - 1u));
        if (gate >= count || inst >= batch) return;
        const GateDesc d = descs[gate];
        eval_one<NR>(d, inst, bstride, W, T, W + (size_t)(out_slot0 + gate) * bstride + inst, rk, te);
        return;
    }
    const uint32_t gate = blockIdx.x / chunks, chunk = blockIdx.x - gate * chunks;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sub = w >> 1, q = w & 1u, lane = threadIdx.x & 63u;
    const uint32_t inst = chunk * 128u + sub * 64u + lane;
    const bool live = inst < batch;
    const size_t i = live ? inst : batch - 1;
    // operands and (the combining wave) table rows first: their round trip runs under the table load and the AES
    const GateDesc d = descs[gate];
    const uint32_t op = d.row_op >> kOpShift;
    const uint4 a = W[(size_t)d.in0 * bstride + i];
    const uint4 b = op == GC_INV ? a : W[(size_t)d.in1 * bstride + i];
    const uint4 *row = T + (size_t)(d.row_op & kRowMask) * bstride + i;
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = make_uint4(0, 0, 0, 0);
    if (q == 0) {
        if (op == GC_OR) {
            const uint32_t index = (lbit_s(a) ? 2u : 0u) | (lbit_s(b) ? 1u : 0u);
            if (index > 0) r0 = row[(size_t)(index - 1) * bstride];
        } else {
            r0 = row[0];
            if (op == GC_AND) r1 = row[bstride];
        }
    }
    load_te_tables(te, g_te0);
    __syncthreads();
    uint4 h[1] = {make_uint4(0, 0, 0, 0)};
    if (op == GC_AND || q == 0) {
        uint32_t k[1][4];
        if (op == GC_OR) make_k(a, b, d.tweak, k[0]);
        else make_k_half(q ? b : a, d.tweak + q, k[0]);
        hash_n<NR, 1>(k, h, rk, te);
    }
    if (q) xh[sub][lane] = h[0];
    __syncthreads();
    if (q != 0 || !live) return;
    uint4 *out = W + (size_t)(out_slot0 + gate) * bstride + inst;
    if (op == GC_AND) {  // eval.go:53-78 (r0 = TG, r1 = TE)
        const uint4 wg = lxor(h[0], land(r0, smask(a)));
        const uint4 we = lxor(xh[sub][lane], land(lxor(r1, a), smask(b)));
        *out = lxor(wg, we);
    } else if (op == GC_INV) {  // eval.go:96-109
        *out = lxor(h[0], land(r0, smask(a)));
    } else {  // GC_OR: eval.go:80-94 (r0 = the row its index names, 0 for index 0)
        *out = lxor(h[0], r0);
    }
}

// GC_LEVEL_WHOLE_GATES=1: the kernels above (a thread garbles / evaluates a whole gate) — the form until round 6, kept for comparison
static bool level_whole_gates() {
    static const bool v = [] {
        const char *e = std::getenv("GC_LEVEL_WHOLE_GATES");
        return e && *e && *e != '0';
    }();
    return v;
}

#define GC_DISPATCH_LEVEL(KERNEL, ...)                                                                     \
    do {                                                                                                   \
        dim3 grid((a.count + (256u >> g.lg) - 1) / (256u >> g.lg), g.yblocks), block(256);                 \
        const bool uni = g.lg >= 6;                                                                        \
        switch (a.rounds) {                                                                                \
        case 10:                                                                                           \
            if (uni) hipLaunchKernelGGL((KERNEL<10, true>), grid, block, 0, s, __VA_ARGS__);               \
            else hipLaunchKernelGGL((KERNEL<10, false>), grid, block, 0, s, __VA_ARGS__);                  \
            break;                                                                                         \
        case 12:                                                                                           \
            if (uni) hipLaunchKernelGGL((KERNEL<12, true>), grid, block, 0, s, __VA_ARGS__);               \
            else hipLaunchKernelGGL((KERNEL<12, false>), grid, block, 0, s, __VA_ARGS__);                  \
            break;                                                                                         \
        default:                                                                                           \
            if (uni) hipLaunchKernelGGL((KERNEL<14, true>), grid, block, 0, s, __VA_ARGS__);               \
            else hipLaunchKernelGGL((KERNEL<14, false>), grid, block, 0, s, __VA_ARGS__);                  \
            break;                                                                                         \
        }                                                                                                  \
    } while (0)

#define GC_DISPATCH_SPLIT(KERNEL, PER, ...)                                                                \
    do {                                                                                                   \
        const uint32_t chunks = (g.batch + (PER) - 1) / (PER), nb_hash = a.nonfree * chunks;               \
        const uint32_t nfree = a.count - a.nonfree, per_blk = 256u >> g.lg;                                \
        const uint32_t gx_free = nfree ? (nfree + per_blk - 1) / per_blk : 1u;                             \
        const dim3 grid(nb_hash + (nfree ? gx_free * g.yblocks : 0u)), block(256);                         \
        switch (a.rounds) {                                                                                \
        case 10: hipLaunchKernelGGL((KERNEL<10>), grid, block, 0, s, __VA_ARGS__, chunks, nb_hash, gx_free); break; \
        case 12: hipLaunchKernelGGL((KERNEL<12>), grid, block, 0, s, __VA_ARGS__, chunks, nb_hash, gx_free); break; \
        default: hipLaunchKernelGGL((KERNEL<14>), grid, block, 0, s, __VA_ARGS__, chunks, nb_hash, gx_free); break; \
        }                                                                                                  \
    } while (0)

void launch_garble_level(const LevelArgs &a, const BatchGeom &g, hipStream_t s) {
    if (a.count == 0) return;
    if (a.nonfree && !level_whole_gates()) {
        GC_DISPATCH_SPLIT(k_garble_level_split, 64u, a.descs, a.count, a.nonfree, a.out_slot0, g.batch, g.bstride, g.lg, a.W, a.R,
                          a.T, a.rk, a.te0);
        return;
    }
    GC_DISPATCH_LEVEL(k_garble_level, a.descs, a.count, a.nonfree, a.out_slot0, g.batch, g.bstride, g.lg, a.W, a.R,
                      a.T, a.rk, a.te0);
}

void launch_eval_level(const LevelArgs &a, const BatchGeom &g, hipStream_t s) {
    if (a.count == 0) return;
    if (a.nonfree && !level_whole_gates()) {
        GC_DISPATCH_SPLIT(k_eval_level_split, 128u, a.descs, a.count, a.nonfree, a.out_slot0, g.batch, g.bstride, g.lg, a.W,
                          (const uint4 *)a.T, a.rk, a.te0);
        return;
    }
    GC_DISPATCH_LEVEL(k_eval_level, a.descs, a.count, a.nonfree, a.out_slot0, g.batch, g.bstride, g.lg, a.W,
                      (const uint4 *)a.T, a.rk, a.te0);
}

// ------------------------------------------------------------------------------------------
// Layout movers: [instance][n] (host / reference order) <-> [slot][instance] (device order).
// 32x32 tiles of 16-byte elements through LDS; both the global read and the global write are
// coalesced along their fast axis.
// ------------------------------------------------------------------------------------------
constexpr int TILE = 32;

__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

// ot.Label.SetData on 16 big-endian bytes loaded as a uint4 of little-endian words:
// D0 = BE(bytes 0..7): D0.hi = bswap(word0), D0.lo = bswap(word1); likewise D1.
__device__ __forceinline__ uint4 label_from_be(uint4 raw) {
    return make_uint4(bswap32(raw.y), bswap32(raw.x), bswap32(raw.w), bswap32(raw.z));
}

// rnd [batch][1+ninputs] -> R [inst], W[w][inst]   (garble.go:253-258, 271-278)
__global__ __launch_bounds__(256) void k_init_garble(const uint4 *__restrict__ rnd, uint32_t ninputs,
                                                     uint4 *__restrict__ W, uint4 *__restrict__ Rv, uint32_t batch,
                                                     Layout lay) {
    __shared__ uint4 tile[TILE][TILE + 1];
    const uint32_t n = ninputs + 1;
    const uint32_t j0 = blockIdx.x * TILE, i0 = blockIdx.y * TILE;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (uint32_t r = ty; r < TILE; r += 8) {
        uint32_t i = i0 + r, j = j0 + tx;
        if (i < batch && j < n) tile[r][tx] = label_from_be(rnd[(size_t)i * n + j]);
    }
    __syncthreads();
    for (uint32_t r = ty; r < TILE; r += 8) {
        uint32_t j = j0 + r, i = i0 + tx;
        if (i < batch && j < n) {
            uint4 v = tile[tx][r];
            if (j == 0) {
                v.y |= 0x80000000u;  // R.SetS(true)
                Rv[i] = v;
            } else {
                W[lay.at(j - 1, i)] = v;
            }
        }
    }
}

void launch_init_garble(const uint4 *rnd, uint32_t ninputs, uint4 *W, uint4 *R, const BatchGeom &g, hipStream_t s) {
    dim3 grid((ninputs + 1 + TILE - 1) / TILE, (g.batch + TILE - 1) / TILE);
    hipLaunchKernelGGL(k_init_garble, grid, dim3(256), 0, s, rnd, ninputs, W, R, g.batch, g.lw);
}

// dst[inst][j] = W[slot(j)][inst]            (mode 0, one label per element)
// dst[inst][j] = {L0, L0 ^ R[inst]}          (mode 1, ot.Wire per element)
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ W, const uint32_t *__restrict__ slots,
                                                uint32_t slot0, uint32_t n, const uint4 *__restrict__ Rv,
                                                uint4 *__restrict__ dst, size_t dst_stride, uint32_t batch,
                                                Layout lay, uint32_t inst0) {
    __shared__ uint4 tile[TILE][TILE + 1];
    const uint32_t j0 = blockIdx.x * TILE, i0 = blockIdx.y * TILE;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t r = ty; r < TILE; r += 8) {
        uint32_t j = j0 + r, i = i0 + tx;
        if (j < n && i < batch) {
            uint32_t slot = slots ? slots[j] : slot0 + j;
            tile[r][tx] = slot == 0xffffffffu ? make_uint4(0, 0, 0, 0) : W[lay.at(slot, inst0 + i)];
        }
    }
    __syncthreads();
    for (uint32_t r = ty; r < TILE; r += 8) {
        uint32_t i = i0 + r, j = j0 + tx;
        if (j < n && i < batch) {
            uint4 v = tile[tx][r];
            if (MODE == 0) {
                dst[(size_t)i * dst_stride + j] = v;
            } else {
                uint4 *o = dst + (size_t)i * dst_stride + 2 * (size_t)j;
                o[0] = v;
                o[1] = lxor(v, Rv[inst0 + i]);
            }
        }
    }
}

void launch_gather(const uint4 *W, const Layout &lay, uint32_t inst0, const uint32_t *slots, uint32_t slot0,
                   uint32_t n, const uint4 *R, int mode, uint4 *dst, size_t dst_stride_elems, uint32_t count,
                   hipStream_t s) {
    if (n == 0 || count == 0) return;
    dim3 grid((n + TILE - 1) / TILE, (count + TILE - 1) / TILE);
    if (mode == 0)
        hipLaunchKernelGGL(k_gather<0>, grid, dim3(256), 0, s, W, slots, slot0, n, R, dst, dst_stride_elems, count, lay,
                           inst0);
    else
        hipLaunchKernelGGL(k_gather<1>, grid, dim3(256), 0, s, W, slots, slot0, n, R, dst, dst_stride_elems, count, lay,
                           inst0);
}

// W[slot(j)][inst] = src[inst][j]
__global__ __launch_bounds__(256) void k_scatter(const uint4 *__restrict__ src, size_t src_stride, uint32_t n,
                                                 const uint32_t *__restrict__ slots, uint32_t slot0,
                                                 uint4 *__restrict__ W, uint32_t batch, Layout lay,
                                                 uint32_t inst0) {
    __shared__ uint4 tile[TILE][TILE + 1];
    const uint32_t j0 = blockIdx.x * TILE, i0 = blockIdx.y * TILE;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t r = ty; r < TILE; r += 8) {
        uint32_t i = i0 + r, j = j0 + tx;
        if (j < n && i < batch) tile[r][tx] = src[(size_t)i * src_stride + j];
    }
    __syncthreads();
    for (uint32_t r = ty; r < TILE; r += 8) {
        uint32_t j = j0 + r, i = i0 + tx;
        if (j < n && i < batch) {
            uint32_t slot = slots ? slots[j] : slot0 + j;
            if (slot != 0xffffffffu) W[lay.at(slot, inst0 + i)] = tile[tx][r];
        }
    }
}

void launch_scatter(const uint4 *src, size_t src_stride_elems, uint32_t n, const uint32_t *slots, uint32_t slot0,
                    uint4 *W, const Layout &lay, uint32_t inst0, uint32_t count, hipStream_t s) {
    if (n == 0 || count == 0) return;
    dim3 grid((n + TILE - 1) / TILE, (count + TILE - 1) / TILE);
    hipLaunchKernelGGL(k_scatter, grid, dim3(256), 0, s, src, src_stride_elems, n, slots, slot0, W, count, lay, inst0);
}

// evaluator's active input labels: L0 ^ (bit ? R : 0)   (LabelForBit, circuit/helpers.go:10-15)
__global__ __launch_bounds__(256) void k_select_inputs(uint4 *__restrict__ We, const uint4 *__restrict__ Wg,
                                                       const uint4 *__restrict__ Rv,
                                                       const uint8_t *__restrict__ bits, uint32_t ninputs,
                                                       uint32_t batch, Layout lay) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= batch) return;
    const uint4 R = Rv[i];
    for (uint32_t w = blockIdx.y; w < ninputs; w += gridDim.y) {
        const uint32_t m = bits[(size_t)i * ninputs + w] ? 0xffffffffu : 0u;
        We[lay.at(w, i)] = lxor(Wg[lay.at(w, i)], land(R, m));
    }
}

void launch_select_inputs(uint4 *We, const uint4 *Wg, const uint4 *R, const uint8_t *bits, uint32_t ninputs,
                          const BatchGeom &g, hipStream_t s) {
    if (ninputs == 0) return;
    dim3 grid((g.batch + 255) / 256, ninputs < 65535u ? ninputs : 65535u);
    hipLaunchKernelGGL(k_select_inputs, grid, dim3(256), 0, s, We, Wg, R, bits, ninputs, g.batch, g.lw);
}

// BitFromLabel (circuit/helpers.go:18-28) over all output wires
__global__ __launch_bounds__(256) void k_decode(const uint4 *__restrict__ Wg, const uint4 *__restrict__ Rv,
                                                const uint4 *__restrict__ We,
                                                const uint32_t *__restrict__ out_slots, uint32_t noutputs,
                                                uint8_t *__restrict__ bits_out, uint32_t *__restrict__ mismatch,
                                                uint32_t batch, Layout lay) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= batch) return;
    const uint4 R = Rv[i];
    for (uint32_t j = blockIdx.y; j < noutputs; j += gridDim.y) {
        const size_t at = lay.at(out_slots[j], i);
        const uint4 l0 = Wg[at], lab = We[at];
        uint8_t bit = 0;
        if (leq(lab, l0)) bit = 0;
        else if (leq(lab, lxor(l0, R))) bit = 1;
        else {
            bit = 0xff;
            if (mismatch) atomicAdd(mismatch, 1u);
        }
        bits_out[(size_t)i * noutputs + j] = bit;
    }
}

void launch_decode(const uint4 *Wg, const uint4 *R, const uint4 *We, const uint32_t *out_slots, uint32_t noutputs,
                   uint8_t *bits_out, uint32_t *mismatch, const BatchGeom &g, hipStream_t s) {
    if (noutputs == 0) return;
    dim3 grid((g.batch + 255) / 256, noutputs < 65535u ? noutputs : 65535u);
    hipLaunchKernelGGL(k_decode, grid, dim3(256), 0, s, Wg, R, We, out_slots, noutputs, bits_out, mismatch, g.batch,
                       g.lw);
}

__global__ __launch_bounds__(256) void k_gather_rows(const uint4 *__restrict__ W, const uint32_t *__restrict__ slots,
                                                     uint32_t n, uint4 *__restrict__ dst, uint32_t batch,
                                                     uint32_t bstride, Layout lay) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= bstride) return;
    for (uint32_t j = blockIdx.y; j < n; j += gridDim.y)
        dst[(size_t)j * bstride + i] = i < batch ? W[lay.at(slots[j], i)] : make_uint4(0, 0, 0, 0);
}

void launch_gather_rows(const uint4 *W, const uint32_t *slots, uint32_t n, uint4 *dst, const BatchGeom &g,
                        hipStream_t s) {
    if (n == 0) return;
    dim3 grid((g.bstride + 255) / 256, n < 65535u ? n : 65535u);
    hipLaunchKernelGGL(k_gather_rows, grid, dim3(256), 0, s, W, slots, n, dst, g.batch, g.bstride, g.lw);
}

// ---- table egress / ingest (wire format of garbler.go:69-82 / evaluator.go:40-66) -----------------------

__device__ __forceinline__ uint32_t rows_of_op(uint32_t op) { return op == GC_AND ? 2u : op == GC_OR ? 3u : op == GC_INV ? 1u : 0u; }

// thread = (gate, instance); consecutive threads of a wave = consecutive gates of one instance, so the 4..52
// bytes each thread writes are adjacent to its neighbours'
__global__ __launch_bounds__(256) void k_tables_egress(const uint4 *__restrict__ T, Layout lt,
                                                       const uint8_t *__restrict__ ops,
                                                       const uint32_t *__restrict__ row_of_gate, uint32_t ngates,
                                                       uint32_t batch, uint8_t *__restrict__ out, size_t stride) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    for (uint32_t inst = blockIdx.y; inst < batch; inst += gridDim.y) {
        uint32_t *base = (uint32_t *)(out + (size_t)inst * stride);
        if (g == 0) base[0] = bswap32(ngates);
        if (g >= ngates) continue;
        const uint32_t row = row_of_gate[g], n = rows_of_op(ops[g]);
        uint32_t *p = base + 1 + g + 4 * (size_t)row;  // byte offset 4 + 4g + 16 row
        p[0] = bswap32(n);
        for (uint32_t r = 0; r < n; r++) {
            const uint4 v = T[lt.at(row + r, inst)];
            p[1 + 4 * r] = bswap32(v.y);  // BE(D0) || BE(D1)
            p[2 + 4 * r] = bswap32(v.x);
            p[3 + 4 * r] = bswap32(v.w);
            p[4 + 4 * r] = bswap32(v.z);
        }
    }
}

__global__ __launch_bounds__(256) void k_tables_ingest(uint4 *__restrict__ T, Layout lt, const uint8_t *__restrict__ ops,
                                                       const uint32_t *__restrict__ row_of_gate, uint32_t ngates,
                                                       uint32_t batch, const uint8_t *__restrict__ in, size_t stride,
                                                       uint32_t *__restrict__ bad) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    for (uint32_t inst = blockIdx.y; inst < batch; inst += gridDim.y) {
        const uint32_t *base = (const uint32_t *)(in + (size_t)inst * stride);
        if (g == 0 && bswap32(base[0]) != ngates) atomicAdd(bad, 1u);
        if (g >= ngates) continue;
        const uint32_t row = row_of_gate[g], n = rows_of_op(ops[g]);
        const uint32_t *p = base + 1 + g + 4 * (size_t)row;
        if (bswap32(p[0]) != n) {
            atomicAdd(bad, 1u);
            continue;
        }
        for (uint32_t r = 0; r < n; r++)
            T[lt.at(row + r, inst)] =
                make_uint4(bswap32(p[2 + 4 * r]), bswap32(p[1 + 4 * r]), bswap32(p[4 + 4 * r]), bswap32(p[3 + 4 * r]));
    }
}

// sha2pc's table encoding (sha2pc/encoding.go:363-411 encodeGarbledTables / :413ff decodeGarbledTables): the rows of
// all gates back to back in gate order, each label as BE(D0) || BE(D1), no headers (garbledTableByteLen =
// 16 * rows).  thread = (row, instance)
__global__ __launch_bounds__(256) void k_slab_be(uint4 *__restrict__ T, Layout lt, uint32_t rows, uint32_t batch,
                                                 uint8_t *__restrict__ buf, size_t stride, bool ingest) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    for (uint32_t inst = blockIdx.y; inst < batch; inst += gridDim.y) {
        uint4 *p = (uint4 *)(buf + (size_t)inst * stride) + r;
        if (ingest) {
            const uint4 v = *p;
            T[lt.at(r, inst)] = make_uint4(bswap32(v.y), bswap32(v.x), bswap32(v.w), bswap32(v.z));
        } else {
            const uint4 v = T[lt.at(r, inst)];
            *p = make_uint4(bswap32(v.y), bswap32(v.x), bswap32(v.w), bswap32(v.z));
        }
    }
}

void launch_tables_egress(const uint4 *T, const Layout &lt, const uint8_t *ops, const uint32_t *row_of_gate,
                          uint32_t ngates, uint32_t batch, uint8_t *out, size_t stride, hipStream_t s) {
    dim3 grid((ngates + 256) / 256, batch < 32768 ? batch : 32768);
    hipLaunchKernelGGL(k_tables_egress, grid, dim3(256), 0, s, T, lt, ops, row_of_gate, ngates, batch, out, stride);
}

void launch_tables_ingest(uint4 *T, const Layout &lt, const uint8_t *ops, const uint32_t *row_of_gate,
                          uint32_t ngates, uint32_t batch, const uint8_t *in, size_t stride, uint32_t *bad,
                          hipStream_t s) {
    dim3 grid((ngates + 256) / 256, batch < 32768 ? batch : 32768);
    hipLaunchKernelGGL(k_tables_ingest, grid, dim3(256), 0, s, T, lt, ops, row_of_gate, ngates, batch, in, stride, bad);
}

void launch_slab_be(uint4 *T, const Layout &lt, uint32_t rows, uint32_t batch, uint8_t *buf, size_t stride, bool ingest,
                    hipStream_t s) {
    if (rows == 0) return;
    dim3 grid((rows + 255) / 256, batch < 32768 ? batch : 32768);
    hipLaunchKernelGGL(k_slab_be, grid, dim3(256), 0, s, T, lt, rows, batch, buf, stride, ingest);
}

// ---- label exchange between a one-instance batch and a device-resident wire store (streaming) ----------------------
// W[slot0 + i] = store[idx[i]]
__global__ void k_store_gather(uint4 *__restrict__ W, const uint4 *__restrict__ store, const uint32_t *__restrict__ idx, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) W[i] = store[idx[i]];
}
// store[idx[j]] = W[slots[j]]; idx 0xffffffff = no store (an output wire that is an input wire is never set)
__global__ void k_store_scatter(uint4 *__restrict__ store, const uint4 *__restrict__ W, const uint32_t *__restrict__ slots,
                                const uint32_t *__restrict__ idx, uint32_t n) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && idx[j] != 0xffffffffu) store[idx[j]] = W[slots[j]];
}
void launch_store_gather(uint4 *W, const uint4 *store, const uint32_t *idx, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_store_gather, dim3((n + 255) / 256), dim3(256), 0, s, W, store, idx, n);
}
void launch_store_scatter(uint4 *store, const uint4 *W, const uint32_t *slots, const uint32_t *idx, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_store_scatter, dim3((n + 255) / 256), dim3(256), 0, s, store, W, slots, idx, n);
}

}  // namespace gc
