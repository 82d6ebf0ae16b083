// stream_serialise.cpp — the device-side serialiser of the streaming garbler: a circuit's gates in the reference's wire
// format (circuit/stream_garble.go:391-446: op + flags byte, 2-3 wire ids as BE u16 / u32, table rows as BE(D0)||BE(D1)),
// byte for byte, written by the GPU at scanned byte offsets (the per-gate host loop was 4 ms for a 131 072-gate step, ten
// times the garbling itself).
#include "stream_internal.h"

namespace gcs {

namespace {
struct SerGate {
    uint32_t ai, bi, ci, size;
    uint8_t op, wc, rows, shortf;
};
__device__ __forceinline__ SerGate ser_gate(const SerArgs &a, uint32_t i) {
    SerGate g{};
    const uint32_t op = a.ops[i];
    const uint32_t w0 = a.gw[3 * i], w1 = a.gw[3 * i + 1], w2 = a.gw[3 * i + 2];
    uint32_t flags = op;
    auto get = [&](uint32_t w, uint32_t bit) -> uint32_t {  // Get/Set indirection (:131-157) + the tmp flag
        if (w < a.first_tmp) return a.in[w];
        if (w >= a.first_out) return a.out[w - a.first_out];
        flags |= bit;
        return w;
    };
    g.bi = op != GC_INV ? get(w1, 0x40) : 0;
    g.ai = get(w0, 0x80);
    g.ci = get(w2, 0x20);
    g.wc = op == GC_INV ? 2 : 3;
    g.rows = op == GC_AND ? 2 : op == GC_OR ? 3 : op == GC_INV ? 1 : 0;
    g.shortf = g.ai <= 0xffff && g.bi <= 0xffff && g.ci <= 0xffff;
    g.op = (uint8_t)(flags | (g.shortf ? 0x10 : 0));
    g.size = 1 + (g.shortf ? 2u : 4u) * g.wc + 16u * g.rows;
    return g;
}
// one gate at p: op byte, 2-3 wire ids (BE u16 if all fit, else BE u32), rows as BE(D0)||BE(D1); T = the dense slab
__device__ __forceinline__ void ser_put(uint8_t *p, const SerGate &q, uint32_t r0, const uint4 *T, const Layout &lt) {
    *p++ = q.op;
    auto put = [&](uint32_t v) {
        if (!q.shortf) {
            *p++ = (uint8_t)(v >> 24);
            *p++ = (uint8_t)(v >> 16);
        }
        *p++ = (uint8_t)(v >> 8);
        *p++ = (uint8_t)v;
    };
    put(q.ai);
    if (q.wc == 3) put(q.bi);
    put(q.ci);
    for (uint32_t r = 0; r < q.rows; r++) {
        const uint4 v = T[lt.at(r0 + r, 0)];  // uint4 label: (x, y) = D0 low / high, (z, w) = D1 low / high
        const uint32_t w4[4] = {v.y, v.x, v.w, v.z};
        for (int j = 0; j < 4; j++) {
            *p++ = (uint8_t)(w4[j] >> 24);
            *p++ = (uint8_t)(w4[j] >> 16);
            *p++ = (uint8_t)(w4[j] >> 8);
            *p++ = (uint8_t)w4[j];
        }
    }
}

// byte size of every block of kSerGates gates
__global__ __launch_bounds__(kSerThreads) void k_ser_sizes(SerArgs a, uint64_t *boff) {
    __shared__ uint32_t red[kSerThreads / 64];
    uint32_t sum = 0;
    for (uint32_t k = 0; k < kSerPer; k++) {
        const uint32_t i = blockIdx.x * kSerGates + threadIdx.x * kSerPer + k;
        if (i < a.ngates) sum += ser_gate(a, i).size;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < kSerThreads / 64; w++) t += red[w];
        boff[blockIdx.x] = t;
    }
}
// exclusive scan of the block sizes in place (one workgroup; a step has a few hundred to a few thousand blocks);
// boff[nblocks] = total
__global__ __launch_bounds__(1024) void k_ser_scan(uint64_t *boff, uint32_t nblocks, uint32_t *total_out = nullptr) {
    __shared__ uint64_t part[1024];
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, nblocks);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += boff[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (uint32_t i = 0; i < 1024; i++) {
            const uint64_t v = part[i];
            part[i] = run;
            run += v;
        }
        boff[nblocks] = run;
        if (total_out) *total_out = (uint32_t)run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; i++) {
        const uint64_t v = boff[i];
        boff[i] = run;
        run += v;
    }
}
// Bytes go out through LDS: a gate is 5 - 61 bytes at an arbitrary byte offset, and byte stores to global memory — every lane of
// a wave in another cache line — kept ONE workgroup busy for 30 us on a 4 000-gate step (the whole of a window-1 step's
// serialiser; profiles/r05b_w1_timeline.txt).  A workgroup builds its contiguous piece of the stream in LDS at the piece's own
// alignment (LDS offset = byte offset mod 16) and writes it out as whole uint4 lines, bytes only at the two ragged ends.
constexpr uint32_t kMaxGateBytes = 1 + 4 * 3 + 16 * 3;  // op, three u32 ids, three rows
__device__ __forceinline__ void flush_piece(const uint8_t *stage, uint8_t *dst, uint32_t sh, uint32_t nbytes, uint32_t nthreads) {
    // stage[sh ..]: the piece; dst: where its first byte goes (dst - sh is 16-byte aligned)
    const uint32_t head = min(nbytes, (16u - sh) & 15u);
    const uint32_t body = (nbytes - head) >> 4, tail = (nbytes - head) & 15u;
    if (threadIdx.x < head) dst[threadIdx.x] = stage[sh + threadIdx.x];
    const uint4 *src16 = (const uint4 *)(stage + sh + head);
    uint4 *dst16 = (uint4 *)(dst + head);
    for (uint32_t k = threadIdx.x; k < body; k += nthreads) dst16[k] = src16[k];
    if (threadIdx.x < tail) dst[head + 16u * body + threadIdx.x] = stage[sh + head + 16u * body + threadIdx.x];
}

// every gate to its byte offset
__global__ __launch_bounds__(kSerThreads) void k_ser_write(SerArgs a, const uint64_t *boff, const uint4 *T, Layout lt,
                                                           uint8_t *buf) {
    __shared__ uint32_t wsum[kSerThreads / 64];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kSerGates * kMaxGateBytes + 16];
    SerGate g[kSerPer];
    uint32_t mine = 0;
    for (uint32_t k = 0; k < kSerPer; k++) {
        const uint32_t i = blockIdx.x * kSerGates + threadIdx.x * kSerPer + k;
        g[k] = SerGate{};
        if (i < a.ngates) g[k] = ser_gate(a, i);
        mine += g[k].size;
    }
    // exclusive scan of the threads' sizes: inside the wave by shuffles, across the waves through LDS
    uint32_t incl = mine;
    const uint32_t lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (uint32_t w = 0; w < kSerThreads / 64; w++) {
        if (w < (threadIdx.x >> 6)) base += wsum[w];
        total += wsum[w];
    }
    const uint64_t b0 = boff[blockIdx.x];
    const uint32_t sh = (uint32_t)b0 & 15u;
    uint32_t pos = sh + base + (incl - mine);
    for (uint32_t k = 0; k < kSerPer; k++) {
        const uint32_t i = blockIdx.x * kSerGates + threadIdx.x * kSerPer + k;
        if (i >= a.ngates) break;
        ser_put(stage + pos, g[k], a.row_of_gate[i], T, lt);
        pos += g[k].size;
    }
    __syncthreads();
    flush_piece(stage, buf + b0, sh, total, kSerThreads);
}

// A step group: workgroup (p, j) = piece p of job j — gates [p * kFinGates, (p + 1) * kFinGates), one per thread.  The byte
// offset of the piece is the size of everything before it, which the workgroup adds up for itself (a job of a group has at
// most kSmallGates gates: 63 sizes per thread for the last piece of the largest); the last piece knows the job's byte count.
constexpr uint32_t kFinGates = 512;
__global__ __launch_bounds__(kFinGates) void k_stream_serialise(const FinJob *jobs) {
    const FinJob j = jobs[blockIdx.y];
    const uint32_t lo = blockIdx.x * kFinGates;
    if (lo >= j.a.ngates && blockIdx.x != 0) return;  // (a job of fewer pieces than the largest of its group)
    __shared__ uint32_t wsum[kFinGates / 64], wpre[kFinGates / 64];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kFinGates * kMaxGateBytes + 16];
    uint32_t before = 0;
    for (uint32_t i = threadIdx.x; i < lo; i += kFinGates) before += ser_gate(j.a, i).size;
    const uint32_t i = lo + threadIdx.x;
    SerGate q{};
    if (i < j.a.ngates) q = ser_gate(j.a, i);
    uint32_t incl = q.size;
    const uint32_t lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o) incl += v;
    }
    for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o, 64);
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    if (lane == 0) wpre[threadIdx.x >> 6] = before;
    __syncthreads();
    uint32_t base = 0, total = 0, b0 = 0;
    for (uint32_t w = 0; w < kFinGates / 64; w++) {
        if (w < (threadIdx.x >> 6)) base += wsum[w];
        total += wsum[w];
        b0 += wpre[w];
    }
    if (threadIdx.x == 0 && lo + kFinGates >= j.a.ngates) *j.size_out = b0 + total;
    const uint32_t sh = b0 & 15u;
    const Layout dense{0, 0, 1, 0};  // one instance: table row r is element r
    if (i < j.a.ngates) ser_put(stage + sh + base + (incl - q.size), q, j.a.row_of_gate[i], j.T, dense);
    __syncthreads();
    flush_piece(stage, j.bytes + b0, sh, total, kFinGates);
}
}  // namespace

void ser_sizes_scan(const SerArgs &a, uint64_t *boff, uint32_t nblocks, uint32_t *total_out, hipStream_t s) {
    hipLaunchKernelGGL(k_ser_sizes, dim3(nblocks), dim3(kSerThreads), 0, s, a, boff);
    hipLaunchKernelGGL(k_ser_scan, dim3(1), dim3(1024), 0, s, boff, nblocks, total_out);
}
void ser_write(const SerArgs &a, const uint64_t *boff, uint32_t nblocks, const uint4 *T, const Layout &lt, uint8_t *buf, hipStream_t s) {
    hipLaunchKernelGGL(k_ser_write, dim3(nblocks), dim3(kSerThreads), 0, s, a, boff, T, lt, buf);
}
void ser_group(const FinJob *d_jobs, uint32_t n, uint32_t max_gates, hipStream_t s) {
    const uint32_t pieces = std::max(1u, (max_gates + kFinGates - 1) / kFinGates);
    hipLaunchKernelGGL(k_stream_serialise, dim3(pieces, n), dim3(kFinGates), 0, s, d_jobs);
}

}  // namespace gcs
