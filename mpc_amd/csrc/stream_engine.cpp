// stream_engine.cpp — C ABI for the streaming garbler (circuit.Streaming, config 5).
//
// Replaces the bodies of NewStreaming (circuit/stream_garble.go:41-75) and Streaming.Garble (:161-192).
// The persistent wire store (stream.wires) stays on the host — it is touched only at circuit
// boundaries; each Garble call (i) resolves the circuit's input wires through in[] (:131-141),
// (ii) garbles the circuit on the device with the SAME kernels as Circuit.Garble (the tweak restarts
// at 0 per circuit, :174, exactly like a fresh Circuit.Garble) and (iii) serialises the gates in the
// reference's wire format (:391-446), byte for byte — ON THE DEVICE: per-gate byte sizes, a two-level exclusive
// scan and a writer kernel that places header and rows at the gate's byte offset; the host copies the finished
// byte string into the caller's buffer (the per-gate host loop was 4 ms for a 131 072-gate step, ten times the
// garbling itself).  Circuits are cached by content, so an SSA instruction that repeats re-uses its levelised
// plan and device buffers.
#include <algorithm>
#include <cstring>
#include <new>
#include <unordered_map>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "engine.h"

using namespace gc;

namespace {
// developer aid: GC_TRACE=1 prints the wall-clock laps of a streaming step to stderr
struct StreamTrace {
    bool on;
    std::chrono::steady_clock::time_point last;
    StreamTrace() : on(std::getenv("GC_TRACE") != nullptr) {
        if (on) last = std::chrono::steady_clock::now();
    }
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gc trace] stream: %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};
}  // namespace

// A cached device circuit with the gate list it was built from.  The key is a 64-bit non-cryptographic hash: a hit is
// only taken after the gates compare equal (an accidental — or, on the evaluator side, peer-crafted — collision would
// otherwise select a plan with other input / output counts: wrong results or out-of-bounds label arrays).
struct CircKey {
    uint32_t in0, in1, out, op;
};
struct CircEntry {
    gc_circ *circ = nullptr;
    std::vector<CircKey> gates;
    uint32_t nwires = 0, nin = 0, nout = 0;
};
using CircCache = std::unordered_multimap<uint64_t, CircEntry>;

struct gc_stream {
    gc_ctx *ctx = nullptr;
    std::vector<uint8_t> key;
    gc_label r{};
    std::vector<gc_label> l0;     // global wire -> L0 (L1 = L0 ^ R)
    std::vector<gc_label> tmp_l0; // stream.tmp (only outputs of the current circuit are meaningful)
    CircCache cache;
    std::vector<uint32_t> alias_gen, alias_j;  // in[] / out[] aliasing check: stamp + index in out[] per global wire
    uint32_t gen = 0;
    std::vector<gc_gate> rewritten;            // gate list with aliased reads redirected (rare)
    uint32_t *d_io = nullptr;   // in[] then out[] of the current call
    size_t io_cap = 0;
    uint64_t *d_boff = nullptr; // per block of kSerGates gates: byte size, then exclusive offset; [nblocks] = total
    size_t boff_cap = 0;
    uint8_t *d_bytes = nullptr; // the serialised circuit
    size_t bytes_cap = 0;
};

namespace {

// ---- device-side serialiser (stream_garble.go:391-446) ---------------------------------------------------
constexpr uint32_t kSerThreads = 256, kSerPer = 4, kSerGates = kSerThreads * kSerPer;

struct SerArgs {
    const uint32_t *gw;   // {in0, in1, out} per gate
    const uint8_t *ops;
    const uint32_t *row_of_gate;
    const uint32_t *in, *out;  // wire maps of this call
    uint32_t ngates, first_tmp, first_out;
};
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
__global__ __launch_bounds__(1024) void k_ser_scan(uint64_t *boff, uint32_t nblocks) {
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
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; i++) {
        const uint64_t v = boff[i];
        boff[i] = run;
        run += v;
    }
}
// every gate to its byte offset: op byte, 2-3 wire ids (BE u16 if all fit, else BE u32), rows as BE(D0)||BE(D1)
__global__ __launch_bounds__(kSerThreads) void k_ser_write(SerArgs a, const uint64_t *boff, const uint4 *T, Layout lt,
                                                           uint8_t *buf) {
    __shared__ uint32_t wsum[kSerThreads / 64];
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
    uint32_t base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += wsum[w];
    size_t pos = boff[blockIdx.x] + base + (incl - mine);
    for (uint32_t k = 0; k < kSerPer; k++) {
        const uint32_t i = blockIdx.x * kSerGates + threadIdx.x * kSerPer + k;
        if (i >= a.ngates) break;
        const SerGate &q = g[k];
        uint8_t *p = buf + pos;
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
        const uint32_t r0 = a.row_of_gate[i];
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
        pos += q.size;
    }
}

template <typename T>
hipError_t grow(T **p, size_t *cap, size_t need) {
    if (need <= *cap) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t n = need + need / 2 + 64;
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e == hipSuccess) *cap = n;
    return e;
}

inline uint64_t be64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}
inline void ensure(gc_stream *s, uint32_t max) {  // ensureWires, 64 Ki-wire pages (:95-100)
    if (max < s->l0.size()) return;
    s->l0.resize(((size_t)max / 0x10000 + 1) * 0x10000, gc_label{0, 0});
}

// content hash of a circuit (cache key): four independent multiply-xor lanes over the gate words, so the
// multiplies of consecutive gates overlap (one dependent chain was 0.2 ms per 131 072-gate step)
uint64_t circuit_hash(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t nin, uint32_t nout) {
    constexpr uint64_t kPrime = 1099511628211ull;
    uint64_t h[4] = {1469598103934665603ull ^ ngates, 0x9e3779b97f4a7c15ull ^ nwires, 0xc2b2ae3d27d4eb4full ^ nin,
                     0x165667b19e3779f9ull ^ nout};
    auto mix = [&](int l, const gc_gate &g) {
        h[l] = (h[l] ^ (((uint64_t)g.in0 << 32) | g.in1)) * kPrime;
        h[l] = (h[l] ^ (((uint64_t)g.out << 8) | g.op)) * kPrime;
    };
    uint32_t i = 0;
    for (; i + 4 <= ngates; i += 4) {
        mix(0, gates[i]);
        mix(1, gates[i + 1]);
        mix(2, gates[i + 2]);
        mix(3, gates[i + 3]);
    }
    for (; i < ngates; i++) mix(i & 3, gates[i]);
    uint64_t r = 0;
    for (int l = 0; l < 4; l++) r = (r ^ h[l]) * kPrime + (r >> 29);
    return r;
}

gc_circ *cache_find(const CircCache &cache, uint64_t h, const gc_gate *gates, uint32_t ngates, uint32_t nwires,
                    uint32_t nin, uint32_t nout) {
    auto range = cache.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
        const CircEntry &e = it->second;
        if (e.gates.size() != ngates || e.nwires != nwires || e.nin != nin || e.nout != nout) continue;
        bool same = true;
        for (uint32_t i = 0; i < ngates && same; i++)
            same = e.gates[i].in0 == gates[i].in0 && e.gates[i].in1 == gates[i].in1 && e.gates[i].out == gates[i].out &&
                   e.gates[i].op == gates[i].op;
        if (same) return e.circ;
    }
    return nullptr;
}

void cache_put(CircCache &cache, uint64_t h, gc_circ *circ, const gc_gate *gates, uint32_t ngates, uint32_t nwires,
               uint32_t nin, uint32_t nout) {
    CircEntry e;
    e.circ = circ;
    e.nwires = nwires, e.nin = nin, e.nout = nout;
    e.gates.resize(ngates);
    for (uint32_t i = 0; i < ngates; i++) e.gates[i] = CircKey{gates[i].in0, gates[i].in1, gates[i].out, gates[i].op};
    cache.emplace(h, std::move(e));
}

}  // namespace

extern "C" {

gc_stream *gc_stream_create(gc_ctx *ctx, const uint8_t *key, size_t keylen, const uint8_t *rnd, size_t rndlen,
                            const uint32_t *inputs, uint32_t ninputs, int *status) try {
    int rc = GC_OK;
    gc_stream *s = nullptr;
    AesKey k;
    if (!ctx || !rnd || (ninputs && !inputs)) rc = GC_E_ARG;
    else if (rndlen < 16) rc = GC_E_RAND;                             // R first (:46)
    else if (!key || !aes_expand_key(key, keylen, &k)) rc = GC_E_KEYSIZE;  // then aes.NewCipher (:52)
    else if (rndlen < 16 * ((size_t)ninputs + 1)) rc = GC_E_RAND;     // then the input labels (:67-73)
    if (rc == GC_OK && !(s = new (std::nothrow) gc_stream)) rc = GC_E_NOMEM;
    if (rc == GC_OK) {
        s->ctx = ctx;
        s->key.assign(key, key + keylen);
        s->r = gc_label{be64(rnd) | 0x8000000000000000ull, be64(rnd + 8)};  // R.SetS(true)
        uint32_t mx = 0;
        for (uint32_t i = 0; i < ninputs; i++) mx = std::max(mx, inputs[i]);
        ensure(s, mx);
        for (uint32_t i = 0; i < ninputs; i++)
            s->l0[inputs[i]] = gc_label{be64(rnd + 16 * ((size_t)i + 1)), be64(rnd + 16 * ((size_t)i + 1) + 8)};
    }
    if (status) *status = rc;
    return s;
} catch (...) {
    const int rc__ = gc::on_exception();
    if (status) *status = rc__;
    return nullptr;
}

void gc_stream_free(gc_stream *s) {
    if (!s) return;
    for (auto &kv : s->cache) gc_circ_free(kv.second.circ);
    if (s->d_io) (void)hipFree(s->d_io);
    if (s->d_boff) (void)hipFree(s->d_boff);
    if (s->d_bytes) (void)hipFree(s->d_bytes);
    delete s;
}

int gc_stream_get_wire(gc_stream *s, uint32_t w, gc_wire *out) {  // Streaming.GetInput (:117-119)
    if (!s || !out || w >= s->l0.size()) return GC_E_ARG;
    out->l0 = s->l0[w];
    out->l1 = gc_label{s->l0[w].d0 ^ s->r.d0, s->l0[w].d1 ^ s->r.d1};
    return GC_OK;
}

int gc_stream_garble(gc_stream *s, const gc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                     uint32_t nin, const uint32_t *out, uint32_t nout, uint8_t *buf, size_t cap, size_t *written) try {
    if (!s || (!gates && ngates) || (nin && !in) || (nout && !out) || !buf || !written) return GC_E_ARG;
    // in[] and out[] may overlap (a circuit whose last wires are input wires): initCircuit (:102-114) takes both as they
    // are, Get / Set resolve a wire through in[] first (:131-157), so such an output id is simply never written
    if (nin > nwires || nout > nwires) return GC_E_ARG;
    const uint32_t first_tmp = nin, first_out = nwires - nout;
    // initCircuit (:102-114)
    uint32_t mx = 0;
    for (uint32_t i = 0; i < nin; i++) mx = std::max(mx, in[i]);
    for (uint32_t i = 0; i < nout; i++) mx = std::max(mx, out[i]);
    ensure(s, mx);
    *written = 0;
    if (ngates == 0) return GC_OK;
    StreamTrace tr;

    // in[] / out[] naming the same GLOBAL wire (wire-id re-use, in-place update): the reference resolves
    // stream.wire(index) per gate (:131-157), so a gate that reads the input-mapped wire after the gate that Set the
    // output-mapped one sees the NEW label.  The device garbles from a snapshot of the inputs: redirect such reads to
    // the producing circuit wire (same global id and flags on the wire, so the serialised bytes do not change).
    {
        if (s->alias_gen.size() < s->l0.size()) {
            s->alias_gen.resize(s->l0.size(), 0);
            s->alias_j.resize(s->l0.size(), 0);
        }
        if (++s->gen == 0) {
            std::fill(s->alias_gen.begin(), s->alias_gen.end(), 0);
            s->gen = 1;
        }
        for (uint32_t j = 0; j < nout; j++)
            if (first_out + j >= first_tmp) {
                s->alias_gen[out[j]] = s->gen;
                s->alias_j[out[j]] = j;
            }
        bool aliased = false;
        for (uint32_t i = 0; i < nin && !aliased; i++) aliased = s->alias_gen[in[i]] == s->gen;
        if (aliased) {
            std::vector<uint8_t> set(nout, 0);
            s->rewritten.assign(gates, gates + ngates);
            for (uint32_t g = 0; g < ngates; g++) {
                gc_gate &q = s->rewritten[g];
                auto redirect = [&](uint32_t w) {
                    if (w < nin && s->alias_gen[in[w]] == s->gen && set[s->alias_j[in[w]]]) return first_out + s->alias_j[in[w]];
                    return w;
                };
                q.in0 = redirect(q.in0);
                if (q.op != GC_INV) q.in1 = redirect(q.in1);
                if (q.out >= first_out && q.out < nwires) set[q.out - first_out] = 1;
            }
            gates = s->rewritten.data();
        }
    }

    // device circuit (cached by content); a new circuit is validated once (garbleGate's checks, :195-210)
    const uint64_t h = circuit_hash(gates, ngates, nwires, nin, nout);
    gc_circ *circ = cache_find(s->cache, h, gates, ngates, nwires, nin, nout);
    if (!circ) {
        for (uint32_t i = 0; i < ngates; i++) {
            if (gates[i].op > GC_INV) return GC_E_GATE;
            if (gates[i].out < first_tmp) return GC_E_ARG;  // a gate writing an input-mapped wire: not produced by the compiler
        }
        int st = GC_OK;
        circ = gc_circ_load(s->ctx, gates, ngates, nwires, nin, nout, &st);
        if (!circ) return st;
        std::vector<uint32_t> gw((size_t)3 * ngates);
        for (uint32_t i = 0; i < ngates; i++) {
            gw[3 * (size_t)i] = gates[i].in0;
            gw[3 * (size_t)i + 1] = gates[i].in1;
            gw[3 * (size_t)i + 2] = gates[i].out;
        }
        hipError_t e = hipMalloc((void **)&circ->d_gwires, gw.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(circ->d_gwires, gw.data(), gw.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            gc_circ_free(circ);
            return GC_E_HIP;
        }
        cache_put(s->cache, h, circ, gates, ngates, nwires, nin, nout);
    }
    tr.lap("alias + hash + cache");
    gc_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t nblocks = (ngates + kSerGates - 1) / kSerGates;

    // (1) byte size of this call's serialisation (depends on in[] / out[]: ids above 0xffff take the long form)
    SerArgs a{};
    uint64_t need = 0;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = grow(&s->d_io, &s->io_cap, (size_t)nin + nout + 1);
        if (e == hipSuccess) e = grow(&s->d_boff, &s->boff_cap, (size_t)nblocks + 1);
        if (e == hipSuccess && nin) e = hipMemcpyAsync(s->d_io, in, nin * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess && nout)
            e = hipMemcpyAsync(s->d_io + nin, out, nout * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return GC_E_HIP;
        a.gw = circ->d_gwires;
        a.ops = circ->d_ops;
        a.row_of_gate = circ->d_row_of_gate;
        a.in = s->d_io;
        a.out = s->d_io + nin;
        a.ngates = ngates;
        a.first_tmp = first_tmp;
        a.first_out = first_out;
        hipLaunchKernelGGL(k_ser_sizes, dim3(nblocks), dim3(kSerThreads), 0, st, a, s->d_boff);
        hipLaunchKernelGGL(k_ser_scan, dim3(1), dim3(1024), 0, st, s->d_boff, nblocks);
        e = hipMemcpyAsync(&need, s->d_boff + nblocks, sizeof(need), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return GC_E_HIP;
    }
    *written = (size_t)need;
    if (need > cap) return GC_E_ARG;
    tr.lap("sizes + scan + sync");

    // (2) input labels through in[] (Get, :131-141); garble; outputs into the global store (Set, :143-157)
    std::vector<gc_label> inl(nin), outl(std::max<uint32_t>(nout, 1));
    for (uint32_t i = 0; i < nin; i++) inl[i] = s->l0[in[i]];
    gc_batch *b = nullptr;
    int rc = gc_garble_labels_keep(circ, s->key.data(), s->key.size(), &s->r, inl.data(), outl.data(), &b);
    if (rc != GC_OK) return rc;
    tr.lap("inputs + garble + outs");
    for (uint32_t j = 0; j < nout; j++)
        if (first_out + j >= first_tmp) s->l0[out[j]] = outl[j];  // an output wire that is an input wire has no gate: no Set

    // (3) wire format (:391-446) written by the device at the scanned offsets, one copy into the caller's buffer
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipError_t e = grow(&s->d_bytes, &s->bytes_cap, (size_t)need);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_ser_write, dim3(nblocks), dim3(kSerThreads), 0, st, a, s->d_boff, b->d_T, b->g.lt, s->d_bytes);
            e = hipMemcpyAsync(buf, s->d_bytes, (size_t)need, hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = GC_E_HIP;
    }
    tr.lap("serialise + d2h");
    gc_circ_release_batch(circ, b);
    return rc;
} catch (...) {
    return gc::on_exception();
}

// ---- streaming evaluator ----------------------------------------------------------------

}  // extern "C"

struct gc_stream_eval {
    gc_ctx *ctx = nullptr;
    std::vector<uint8_t> key;
    std::vector<gc_label> wires;  // StreamEval.wires (global store)
    CircCache cache;
    // per-circuit scratch, kept across calls: last writer of every tmp / global wire with a generation stamp
    std::vector<uint32_t> cur_t, stamp_t, cur_w, stamp_w;
    uint32_t gen = 0;
    std::vector<gc_gate> gates;
    std::vector<gc_label> slab;
};

extern "C" {

gc_stream_eval *gc_stream_eval_create(gc_ctx *ctx, const uint8_t *key, size_t keylen, int *status) try {
    int rc = GC_OK;
    AesKey k;
    gc_stream_eval *e = nullptr;
    if (!ctx) rc = GC_E_ARG;
    else if (!key || !aes_expand_key(key, keylen, &k)) rc = GC_E_KEYSIZE;
    else if (!(e = new (std::nothrow) gc_stream_eval)) rc = GC_E_NOMEM;
    if (e) {
        e->ctx = ctx;
        e->key.assign(key, key + keylen);
    }
    if (status) *status = rc;
    return e;
} catch (...) {
    const int rc__ = gc::on_exception();
    if (status) *status = rc__;
    return nullptr;
}

void gc_stream_eval_free(gc_stream_eval *e) {
    if (!e) return;
    for (auto &kv : e->cache) gc_circ_free(kv.second.circ);
    delete e;
}

int gc_stream_eval_set_wire(gc_stream_eval *e, uint32_t w, const gc_label *l) try {
    if (!e || !l) return GC_E_ARG;
    if (w >= e->wires.size()) e->wires.resize((size_t)w + 1, gc_label{0, 0});
    e->wires[w] = *l;
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_stream_eval_get_wire(gc_stream_eval *e, uint32_t w, gc_label *l) {
    if (!e || !l || w >= e->wires.size()) return GC_E_ARG;
    *l = e->wires[w];
    return GC_OK;
}

int gc_stream_eval_circuit(gc_stream_eval *e, uint32_t ngates, uint32_t ntmp, uint32_t nwires, const uint8_t *buf,
                           size_t len, size_t *consumed) try {
    if (!e || (!buf && len) || !consumed) return GC_E_ARG;
    *consumed = 0;
    // The block comes from the peer: bound it before anything is sized by it.  A gate is at least 5 bytes (op + two
    // 16-bit ids), so a header that announces more gates than the bytes can hold is a truncated stream; global wire
    // ids must stay below the numWires of the block's own header (the reference indexes its store with them,
    // stream_evaluator.go:29-96: an id beyond it panics there) and tmp ids below numTmpWires.
    if ((size_t)ngates > len / 5) return GC_E_ROWS;
    if (e->wires.size() < nwires) e->wires.resize(nwires, gc_label{0, 0});  // InitCircuit(numWires, numTmpWires)
    if (ngates == 0) return GC_OK;
    // Parse the gate stream (stream_evaluator.go:272-345) into an SSA gate list.  Wire ids of the device circuit:
    //   [0, nin)            wires read before this circuit writes them, in order of first use
    //   then one id per gate: first the gates that write a tmp wire, last the gates that write a global wire —
    //   those are the circuit's outputs and the only labels that come back (streaming.Set, :346-432).
    // "Who wrote (tmp, idx) last" is an array look-up with a generation stamp (no clearing between circuits).
    // A tmp wire is private to its OpCircuit block (stream_garble.go:131-157 gives every non-input, non-output
    // wire of the circuit a tmp id, written by a gate before any gate reads it): a block that reads a tmp it has
    // not written is rejected instead of evaluated on a stale label.
    struct Ref { bool tmp; uint32_t idx; };
    if (e->stamp_t.size() < ntmp) {
        e->stamp_t.resize(ntmp, 0);
        e->cur_t.resize(ntmp, 0);
    }
    if (++e->gen == 0) {  // stamp wrap-around
        std::fill(e->stamp_t.begin(), e->stamp_t.end(), 0);
        std::fill(e->stamp_w.begin(), e->stamp_w.end(), 0);
        e->gen = 1;
    }
    const uint32_t gen = e->gen;
    std::vector<gc_gate> &gates = e->gates;
    std::vector<gc_label> &slab = e->slab;
    gates.resize(ngates);
    slab.clear();
    std::vector<Ref> dst(ngates), inputs;
    uint32_t n_global = 0;
    size_t pos = 0;
    for (uint32_t g = 0; g < ngates; g++) {
        if (pos + 1 > len) return GC_E_ROWS;
        uint8_t gop = buf[pos++];
        const bool at = gop & 0x80, bt = gop & 0x40, ct = gop & 0x20, shortf = gop & 0x10;
        gop &= 0x0f;
        if (gop > GC_INV) return GC_E_GATE;  // "invalid operation"
        const int nw = gop == GC_INV ? 2 : 3;
        const size_t sz = shortf ? 2 : 4;
        const uint32_t rows = gop == GC_AND ? 2 : gop == GC_OR ? 3 : gop == GC_INV ? 1 : 0;
        if (pos + sz * nw + 16 * (size_t)rows > len) return GC_E_ROWS;
        uint32_t w[3] = {0, 0, 0};
        for (int i = 0; i < nw; i++) {
            for (size_t b = 0; b < sz; b++) w[i] = (w[i] << 8) | buf[pos + b];
            pos += sz;
        }
        for (uint32_t r = 0; r < rows; r++) {
            slab.push_back(gc_label{be64(buf + pos), be64(buf + pos + 8)});
            pos += 16;
        }
        int err = GC_OK;
        auto use = [&](bool t, uint32_t idx) -> uint32_t {  // current id of a wire; bit 31: a circuit input
            if (t) {
                if (idx >= ntmp || e->stamp_t[idx] != gen) {
                    err = GC_E_ARG;
                    return 0;
                }
                return e->cur_t[idx];
            }
            if (idx >= nwires) {
                err = GC_E_ARG;
                return 0;
            }
            if (idx >= e->stamp_w.size()) {
                e->stamp_w.resize((size_t)idx + 1 + e->stamp_w.size() / 2, 0);
                e->cur_w.resize(e->stamp_w.size(), 0);
            }
            if (e->stamp_w[idx] == gen) return e->cur_w[idx];
            const uint32_t id = 0x80000000u | (uint32_t)inputs.size();
            inputs.push_back(Ref{false, idx});
            e->stamp_w[idx] = gen;
            e->cur_w[idx] = id;
            return id;
        };
        gates[g].in0 = use(at, w[0]);
        gates[g].in1 = nw == 3 ? use(bt, w[1]) : gates[g].in0;
        if (err != GC_OK) return err;
        gates[g].op = gop;
        gates[g].level = 0;
        const uint32_t ci = w[nw - 1];
        dst[g] = Ref{ct, ci};
        if (ct) {
            if (ci >= ntmp) return GC_E_ARG;
            e->stamp_t[ci] = gen;
            e->cur_t[ci] = g;
        } else {
            if (ci >= nwires) return GC_E_ARG;
            if (ci >= e->stamp_w.size()) {
                e->stamp_w.resize((size_t)ci + 1 + e->stamp_w.size() / 2, 0);
                e->cur_w.resize(e->stamp_w.size(), 0);
            }
            e->stamp_w[ci] = gen;
            e->cur_w[ci] = g;
            n_global++;
        }
        gates[g].out = g;  // gate index for now; numbered below
    }
    const uint32_t nin = (uint32_t)inputs.size(), nout = n_global, n_tmp = ngates - n_global;
    {
        std::vector<uint32_t> id_of(ngates);
        uint32_t kt = 0, kg = 0;
        for (uint32_t g = 0; g < ngates; g++) id_of[g] = dst[g].tmp ? nin + kt++ : nin + n_tmp + kg++;
        for (uint32_t g = 0; g < ngates; g++) {
            auto fix = [&](uint32_t v) { return (v & 0x80000000u) ? (v & 0x7fffffffu) : id_of[v]; };
            gates[g].in0 = fix(gates[g].in0);
            gates[g].in1 = gates[g].op == GC_INV ? 0 : fix(gates[g].in1);
            gates[g].out = id_of[g];
        }
    }
    const uint32_t cw = nin + ngates;
    // device circuit, cached by content
    const uint64_t h = circuit_hash(gates.data(), ngates, cw, nin, nout);
    gc_circ *circ = cache_find(e->cache, h, gates.data(), ngates, cw, nin, nout);
    if (!circ) {
        int st = GC_OK;
        circ = gc_circ_load(e->ctx, gates.data(), ngates, cw, nin, nout, &st);
        if (!circ) return st;
        cache_put(e->cache, h, circ, gates.data(), ngates, cw, nin, nout);
    }
    std::vector<gc_label> inl(std::max<uint32_t>(nin, 1)), outl(std::max<uint32_t>(nout, 1));
    for (uint32_t i = 0; i < nin; i++) {
        const Ref &r = inputs[i];
        if (r.idx >= e->wires.size()) e->wires.resize((size_t)r.idx + 1, gc_label{0, 0});
        inl[i] = e->wires[r.idx];
    }
    int rc = gc_eval(circ, e->key.data(), e->key.size(), 1, nullptr, inl.data(), slab.data(), slab.size(), outl.data());
    if (rc != GC_OK) return rc;
    uint32_t k = 0;
    for (uint32_t g = 0; g < ngates; g++) {  // streaming.Set(cTmp, cIndex, output) of the global wires, in gate order
        const Ref &d = dst[g];
        if (d.tmp) continue;
        if (d.idx >= e->wires.size()) e->wires.resize((size_t)d.idx + 1, gc_label{0, 0});
        e->wires[d.idx] = outl[k++];
    }
    *consumed = pos;
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

}  // extern "C"
