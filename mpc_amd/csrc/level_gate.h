// level_gate.h — ONE gate of ONE instance, the plain per-gate form of the reference's loops (circuit/garble.go:311-482,
// circuit/eval.go:28-112) on the classic 4 x 1 KiB T-tables: the body of the level-launch kernels (gc_kernels.hip, schedule 0)
// and of the stand-by that does a cooperative one-instance pass again when it lost a workgroup (fused_kernels.hip).
#pragma once

#include "aes_device.h"
#include "kernels.h"

namespace gc {

// one gate of one instance (garble.go:311-482): d = its descriptor, i = the instance's column, out = where its zero label goes
template <int NR>
__device__ __forceinline__ void garble_one(const GateDesc d, size_t i, uint32_t bstride, uint4 *__restrict__ W, const uint4 R,
                                           uint4 *__restrict__ T, uint4 *out, const uint32_t *__restrict__ rk, const uint32_t *te) {
    const uint32_t op = d.row_op >> kOpShift;
    const uint4 a0 = W[(size_t)d.in0 * bstride + i];

    if (op == GC_XOR) {  // garble.go:331-340
        *out = lxor(a0, W[(size_t)d.in1 * bstride + i]);
        return;
    }
    if (op == GC_XNOR) {  // garble.go:342-351: L0 = a0^b0^R
        *out = lxor(lxor(a0, W[(size_t)d.in1 * bstride + i]), R);
        return;
    }
    uint4 *row = T + (size_t)(d.row_op & kRowMask) * bstride + i;
    if (op == GC_AND) {  // garble.go:353-395
        const uint4 b0 = W[(size_t)d.in1 * bstride + i];
        const uint4 a1 = lxor(a0, R), b1 = lxor(b0, R);
        uint32_t k[4][4];
        make_k_half(a0, d.tweak, k[0]);
        make_k_half(a1, d.tweak, k[1]);
        make_k_half(b0, d.tweak + 1, k[2]);
        make_k_half(b1, d.tweak + 1, k[3]);
        uint4 h[4];
        hash_n<NR, 4>(k, h, rk, te);
        const uint32_t pa = smask(a0), pb = smask(b0);
        // first half gate: TG = H(a0)^H(a1)^(pb?R:0); WG0 = H(a0)^(pa?TG:0)
        uint4 tg = lxor(lxor(h[0], h[1]), land(R, pb));
        uint4 wg0 = lxor(h[0], land(tg, pa));
        // second half gate: TE = H(b0)^H(b1)^a0; WE0 = H(b0)^(pb?TE^a0:0)
        uint4 te_ = lxor(lxor(h[2], h[3]), a0);
        uint4 we0 = lxor(h[2], land(lxor(te_, a0), pb));
        *out = lxor(wg0, we0);
        row[0] = tg;
        row[bstride] = te_;
        return;
    }
    if (op == GC_INV) {  // garble.go:446-474; enc(a,0,0,id) == H(a,id)
        const uint4 a1 = lxor(a0, R);
        uint32_t k[2][4];
        make_k_half(a0, d.tweak, k[0]);
        make_k_half(a1, d.tweak, k[1]);
        uint4 e[2];
        hash_n<NR, 2>(k, e, rk, te);
        // table[S(a0)] = E0, table[S(a1)] = E1; row 0 is folded into the output labels:
        // S(a0)==0: L0 = E0^R (L1 = E0);  S(a0)==1: L0 = E1 (L1 = E1^R);  row[1] = E0^E1^R
        const bool s = lbit_s(a0);
        *out = s ? e[1] : lxor(e[0], R);
        row[0] = lxor(lxor(e[0], e[1]), R);
        return;
    }
    // GC_OR: garble.go:412-444
    {
        const uint4 b0 = W[(size_t)d.in1 * bstride + i];
        const uint4 a1 = lxor(a0, R), b1 = lxor(b0, R);
        uint32_t k[4][4];
        make_k(a0, b0, d.tweak, k[0]);
        make_k(a0, b1, d.tweak, k[1]);
        make_k(a1, b0, d.tweak, k[2]);
        make_k(a1, b1, d.tweak, k[3]);
        uint4 e[4];  // e[2u+v] = enc(a_u, b_v, 0, id)
        hash_n<NR, 4>(k, e, rk, te);
        const uint32_t pa = lbit_s(a0) ? 1u : 0u, pb = lbit_s(b0) ? 1u : 0u;
        const uint32_t l0 = 2u * pa + pb;  // idx(a0,b0)
        // table[2s+t] = e[2(s^pa) + (t^pb)]
        // (mask selects keep everything in registers — no dynamically indexed arrays)
        const uint32_t m0 = l0 == 0 ? ~0u : 0u, m1 = l0 == 1 ? ~0u : 0u, m2 = l0 == 2 ? ~0u : 0u,
                       m3 = l0 == 3 ? ~0u : 0u;
        auto pick = [&](uint32_t ma, uint32_t mb, uint32_t mc, uint32_t md) {
            // e[q ^ l0] for the q whose (q^l0) pattern is given by the masks of l0
            return lxor(lxor(land(e[0], ma), land(e[1], mb)), lxor(land(e[2], mc), land(e[3], md)));
        };
        const uint4 t0 = pick(m0, m1, m2, m3);  // q=0: src = l0
        const uint4 t1 = pick(m1, m0, m3, m2);  // q=1: src = l0^1
        const uint4 t2 = pick(m2, m3, m0, m1);  // q=2: src = l0^2
        const uint4 t3 = pick(m3, m2, m1, m0);  // q=3: src = l0^3
        // c.L0 = c.L1 = table[0]; the label of the (a0,b0) row gets no R when l0 == 0
        const uint4 c0 = lxor(t0, land(R, ~m0));
        const uint4 c1 = lxor(t0, land(R, m0));
        // table[i] ^= (i == l0) ? c.L0 : c.L1, rows 1..3 are emitted
        row[0] = lxor(t1, lxor(land(c0, m1), land(c1, ~m1)));
        row[bstride] = lxor(t2, lxor(land(c0, m2), land(c1, ~m2)));
        row[2 * (size_t)bstride] = lxor(t3, lxor(land(c0, m3), land(c1, ~m3)));
        *out = c0;
    }
}

// one gate of one instance (eval.go:28-112)
template <int NR>
__device__ __forceinline__ void eval_one(const GateDesc d, size_t i, uint32_t bstride, uint4 *__restrict__ W,
                                         const uint4 *__restrict__ T, uint4 *out, const uint32_t *__restrict__ rk, const uint32_t *te) {
    const uint32_t op = d.row_op >> kOpShift;
    const uint4 a = W[(size_t)d.in0 * bstride + i];

    if (op == GC_XOR || op == GC_XNOR) {  // eval.go:49-51
        *out = lxor(a, W[(size_t)d.in1 * bstride + i]);
        return;
    }
    const uint4 *row = T + (size_t)(d.row_op & kRowMask) * bstride + i;
    if (op == GC_AND) {  // eval.go:53-78
        const uint4 b = W[(size_t)d.in1 * bstride + i];
        const uint4 tg = row[0], te_ = row[bstride];
        uint32_t k[2][4];
        make_k_half(a, d.tweak, k[0]);
        make_k_half(b, d.tweak + 1, k[1]);
        uint4 h[2];
        hash_n<NR, 2>(k, h, rk, te);
        uint4 wg = lxor(h[0], land(tg, smask(a)));
        uint4 we = lxor(h[1], land(lxor(te_, a), smask(b)));
        *out = lxor(wg, we);
        return;
    }
    if (op == GC_INV) {  // eval.go:96-109: c = S(a) ? row[0] : 0; out = c ^ pi(K) ^ K
        uint32_t k[1][4];
        make_k_half(a, d.tweak, k[0]);
        uint4 h[1];
        hash_n<NR, 1>(k, h, rk, te);
        *out = lxor(h[0], land(row[0], smask(a)));
        return;
    }
    {  // GC_OR: eval.go:80-94
        const uint4 b = W[(size_t)d.in1 * bstride + i];
        const uint32_t index = (lbit_s(a) ? 2u : 0u) | (lbit_s(b) ? 1u : 0u);
        uint4 c = make_uint4(0, 0, 0, 0);
        if (index > 0) c = row[(size_t)(index - 1) * bstride];
        uint32_t k[1][4];
        make_k(a, b, d.tweak, k[0]);
        uint4 h[1];
        hash_n<NR, 1>(k, h, rk, te);
        *out = lxor(h[0], c);
    }
}

}  // namespace gc
