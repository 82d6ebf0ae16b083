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

// Global wire store of a stream (Streaming.wires / StreamEval.wires, stream_garble.go:27-38, stream_evaluator.go:29-34) in
// HBM: the circuits of consecutive steps hand labels to each other on the device (gather before / scatter after every
// pass), so a step does not wait for the GPU at all and the host work of step k + 1 (hashing, cache look-up, parsing)
// overlaps the kernels of step k.  Labels the HOST sets (the stream's inputs) live in a host shadow until the next pass
// uploads them; labels a circuit wrote are read back on demand (GetInput / OpReturn are rare).
struct DevStore {
    std::vector<gc_label> host;   // valid where !on_dev
    std::vector<uint8_t> on_dev;  // 1: the current label was written by a circuit on the device
    std::vector<uint32_t> dirty;  // host-set wires not uploaded yet
    uint4 *d = nullptr;
    size_t cap = 0;

    void ensure(size_t n) {
        if (host.size() < n) {
            host.resize(n, gc_label{0, 0});
            on_dev.resize(n, 0);
        }
    }
    void set(uint32_t w, const gc_label &l) {
        ensure((size_t)w + 1);
        host[w] = l;
        on_dev[w] = 0;
        dirty.push_back(w);
    }
    // device array covers every wire, host-set labels are uploaded (runs of consecutive indices as one copy)
    int flush(gc_ctx *ctx) {
        hipStream_t st = ctx->stream;
        if (host.size() > cap) {
            const size_t ncap = std::max(host.size(), cap * 2);
            uint4 *nd = nullptr;
            GC_HIP(hipMalloc((void **)&nd, ncap * sizeof(uint4)));
            GC_HIP(hipMemsetAsync(nd, 0, ncap * sizeof(uint4), st));  // a never-set wire reads as the zero label
            if (d) GC_HIP(hipMemcpyAsync(nd, d, cap * sizeof(uint4), hipMemcpyDeviceToDevice, st));
            GC_HIP(hipStreamSynchronize(st));
            if (d) (void)hipFree(d);
            d = nd;
            cap = ncap;
        }
        for (size_t i = 0; i < dirty.size();) {
            size_t j = i + 1;
            while (j < dirty.size() && dirty[j] == dirty[j - 1] + 1) j++;
            // a wire set twice keeps its last value in host[]; one that a circuit overwrote meanwhile is skipped
            bool all_host = true;
            for (size_t k = i; k < j; k++) all_host = all_host && !on_dev[dirty[k]];
            if (all_host)
                GC_HIP(hipMemcpyAsync(d + dirty[i], &host[dirty[i]], (j - i) * sizeof(gc_label), hipMemcpyHostToDevice, st));
            else
                for (size_t k = i; k < j; k++)
                    if (!on_dev[dirty[k]])
                        GC_HIP(hipMemcpyAsync(d + dirty[k], &host[dirty[k]], sizeof(gc_label), hipMemcpyHostToDevice, st));
            i = j;
        }
        dirty.clear();
        return GC_OK;
    }
    int get(gc_ctx *ctx, uint32_t w, gc_label *out) {
        if (w >= host.size()) return GC_E_ARG;
        if (!on_dev[w]) {
            *out = host[w];
            return GC_OK;
        }
        GC_HIP(hipSetDevice(ctx->device));
        GC_HIP(hipMemcpyAsync(out, d + w, sizeof(gc_label), hipMemcpyDeviceToHost, ctx->stream));
        GC_HIP(hipStreamSynchronize(ctx->stream));
        host[w] = *out;  // cache: unchanged until a circuit writes the wire again
        on_dev[w] = 0;
        return GC_OK;
    }
    void release() {
        if (d) (void)hipFree(d);
        d = nullptr;
        cap = 0;
    }
};

struct gc_stream {
    gc_ctx *ctx = nullptr;
    std::vector<uint8_t> key;
    gc_label r{};
    DevStore store;               // global wire -> L0 (L1 = L0 ^ R)
    CircCache cache;
    std::vector<uint32_t> alias_gen, alias_j;  // in[] / out[] aliasing check: stamp + index in out[] per global wire
    uint32_t gen = 0;
    std::vector<gc_gate> rewritten;            // gate list with aliased reads redirected (rare)
    std::vector<uint32_t> io_host;             // in[], out[], out[] with 0xffffffff where nothing is stored
    uint32_t *d_io = nullptr;   // device copy of io_host for the call in flight
    size_t io_cap = 0;
    uint64_t *d_boff = nullptr; // per block of kSerGates gates: byte size, then exclusive offset; [nblocks] = total
    size_t boff_cap = 0;
    // two steps may be in flight (gc_stream_garble_begin / _finish): their serialised bytes, sizes and completion events
    uint8_t *d_bytes[2] = {nullptr, nullptr};
    size_t bytes_cap[2] = {0, 0};
    uint64_t *need_host = nullptr;  // pinned [2]
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    uint32_t head = 0, pending = 0;  // slot of the oldest step in flight, steps in flight
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
    if (max < s->store.host.size()) return;
    s->store.ensure(((size_t)max / 0x10000 + 1) * 0x10000);
}

// content hash of a circuit (cache key): four independent multiply-xor lanes over the gate words, so the
// multiplies of consecutive gates overlap (one dependent chain was 0.2 ms per 131 072-gate step)
struct CircuitHash {  // incremental form: the evaluator hashes while it renumbers (one pass over the gates less)
    static constexpr uint64_t kPrime = 1099511628211ull;
    uint64_t h[4];
    CircuitHash(uint32_t ngates, uint32_t nwires, uint32_t nin, uint32_t nout)
        : h{1469598103934665603ull ^ ngates, 0x9e3779b97f4a7c15ull ^ nwires, 0xc2b2ae3d27d4eb4full ^ nin,
            0x165667b19e3779f9ull ^ nout} {}
    inline void mix(uint32_t i, const gc_gate &g) {
        uint64_t &v = h[i & 3];
        v = (v ^ (((uint64_t)g.in0 << 32) | g.in1)) * kPrime;
        v = (v ^ (((uint64_t)g.out << 8) | g.op)) * kPrime;
    }
    inline void mix(uint32_t i, const CircKey &g) {
        uint64_t &v = h[i & 3];
        v = (v ^ (((uint64_t)g.in0 << 32) | g.in1)) * kPrime;
        v = (v ^ (((uint64_t)g.out << 8) | g.op)) * kPrime;
    }
    uint64_t done() const {
        uint64_t r = 0;
        for (int l = 0; l < 4; l++) r = (r ^ h[l]) * kPrime + (r >> 29);
        return r;
    }
};
uint64_t circuit_hash(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t nin, uint32_t nout) {
    CircuitHash ch(ngates, nwires, nin, nout);
    uint32_t i = 0;
    for (; i + 4 <= ngates; i += 4) {  // four independent lanes: the multiplies of consecutive gates overlap
        ch.mix(0, gates[i]);
        ch.mix(1, gates[i + 1]);
        ch.mix(2, gates[i + 2]);
        ch.mix(3, gates[i + 3]);
    }
    for (; i < ngates; i++) ch.mix(i, gates[i]);
    return ch.done();
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

// the same on the evaluator's packed gate records (16 bytes, no padding: one memcmp)
gc_circ *cache_find_keys(const CircCache &cache, uint64_t h, const std::vector<CircKey> &keys, uint32_t nwires, uint32_t nin,
                         uint32_t nout) {
    static_assert(sizeof(CircKey) == 16, "CircKey must be four packed words");
    auto range = cache.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
        const CircEntry &e = it->second;
        if (e.gates.size() != keys.size() || e.nwires != nwires || e.nin != nin || e.nout != nout) continue;
        if (std::memcmp(e.gates.data(), keys.data(), keys.size() * sizeof(CircKey)) == 0) return e.circ;
    }
    return nullptr;
}

void cache_put_keys(CircCache &cache, uint64_t h, gc_circ *circ, const std::vector<CircKey> &keys, uint32_t nwires,
                    uint32_t nin, uint32_t nout) {
    CircEntry e;
    e.circ = circ;
    e.nwires = nwires, e.nin = nin, e.nout = nout;
    e.gates = keys;
    cache.emplace(h, std::move(e));
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
            s->store.set(inputs[i], gc_label{be64(rnd + 16 * ((size_t)i + 1)), be64(rnd + 16 * ((size_t)i + 1) + 8)});
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->need_host, 2 * sizeof(uint64_t), hipHostMallocDefault);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&s->done[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            set_error("gc_stream_create", e);
            rc = GC_E_HIP;
            gc_stream_free(s);
            s = nullptr;
        }
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
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
    }
    if (s->copy_stream) {
        (void)hipStreamSynchronize(s->copy_stream);
        (void)hipStreamDestroy(s->copy_stream);
    }
    for (auto &kv : s->cache) gc_circ_free(kv.second.circ);
    if (s->d_io) (void)hipFree(s->d_io);
    if (s->d_boff) (void)hipFree(s->d_boff);
    for (int i = 0; i < 2; i++) {
        if (s->d_bytes[i]) (void)hipFree(s->d_bytes[i]);
        if (s->done[i]) (void)hipEventDestroy(s->done[i]);
    }
    if (s->need_host) (void)hipHostFree(s->need_host);
    s->store.release();
    delete s;
}

int gc_stream_get_wire(gc_stream *s, uint32_t w, gc_wire *out) {  // Streaming.GetInput (:117-119)
    if (!s || !out) return GC_E_ARG;
    gc_label l0;
    int rc = s->store.get(s->ctx, w, &l0);
    if (rc != GC_OK) return rc;
    out->l0 = l0;
    out->l1 = gc_label{l0.d0 ^ s->r.d0, l0.d1 ^ s->r.d1};
    return GC_OK;
}

// Streaming.Garble in two halves (additive): _begin enqueues everything for one circuit — the sizes of its serialisation,
// the gather of its input labels from the device-resident wire store, the garbling, the scatter of its output labels
// and the serialiser — and returns WITHOUT waiting; _finish hands out the bytes of the oldest circuit in flight.  With
// begin(k + 1) before finish(k) the host's share of a step (content hash, cache look-up, launches) overlaps the GPU's
// share of the step before: the bytes still leave in order.  At most two circuits in flight.
int gc_stream_garble_begin(gc_stream *s, const gc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                           uint32_t nin, const uint32_t *out, uint32_t nout) try {
    if (!s || (!gates && ngates) || (nin && !in) || (nout && !out)) return GC_E_ARG;
    if (s->pending >= 2) return GC_E_ARG;
    // in[] and out[] may overlap (a circuit whose last wires are input wires): initCircuit (:102-114) takes both as they
    // are, Get / Set resolve a wire through in[] first (:131-157), so such an output id is simply never written
    if (nin > nwires || nout > nwires) return GC_E_ARG;
    const uint32_t first_tmp = nin, first_out = nwires - nout;
    // initCircuit (:102-114)
    uint32_t mx = 0;
    for (uint32_t i = 0; i < nin; i++) mx = std::max(mx, in[i]);
    for (uint32_t i = 0; i < nout; i++) mx = std::max(mx, out[i]);
    ensure(s, mx);
    const uint32_t slot = (s->head + s->pending) & 1u;
    gc_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    GC_HIP(hipSetDevice(ctx->device));
    if (ngates == 0) {  // nothing on the wire
        s->need_host[slot] = 0;
        GC_HIP(hipEventRecord(s->done[slot], st));
        s->pending++;
        return GC_OK;
    }
    StreamTrace tr;

    // in[] / out[] naming the same GLOBAL wire (wire-id re-use, in-place update): the reference resolves
    // stream.wire(index) per gate (:131-157), so a gate that reads the input-mapped wire after the gate that Set the
    // output-mapped one sees the NEW label.  The device garbles from a snapshot of the inputs: redirect such reads to
    // the producing circuit wire (same global id and flags on the wire, so the serialised bytes do not change).
    {
        if (s->alias_gen.size() < s->store.host.size()) {
            s->alias_gen.resize(s->store.host.size(), 0);
            s->alias_j.resize(s->store.host.size(), 0);
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
        int stc = GC_OK;
        circ = gc_circ_load(s->ctx, gates, ngates, nwires, nin, nout, &stc);
        if (!circ) return stc;
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
    const uint32_t nblocks = (ngates + kSerGates - 1) / kSerGates;

    // (1) this call's wire maps on the device: in[], out[], and out[] with "no store" marks for the scatter (an output
    //     wire that is an input wire has no gate: no Set); host-set labels of the store are uploaded
    s->io_host.resize((size_t)nin + 2 * (size_t)nout + 1);
    for (uint32_t i = 0; i < nin; i++) s->io_host[i] = in[i];
    for (uint32_t j = 0; j < nout; j++) {
        s->io_host[nin + j] = out[j];
        s->io_host[nin + nout + j] = first_out + j >= first_tmp ? out[j] : 0xffffffffu;
    }
    SerArgs a{};
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        int rcs = s->store.flush(ctx);
        if (rcs != GC_OK) return rcs;
        hipError_t e = grow(&s->d_io, &s->io_cap, s->io_host.size());
        if (e == hipSuccess) e = grow(&s->d_boff, &s->boff_cap, (size_t)nblocks + 1);
        // the bytes of this step: 13 header bytes + 3 rows per gate at most
        if (e == hipSuccess) e = grow(&s->d_bytes[slot], &s->bytes_cap[slot], (size_t)ngates * 61 + 16);
        if (e == hipSuccess)
            e = hipMemcpyAsync(s->d_io, s->io_host.data(), s->io_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            set_error("gc_stream_garble", e);
            return e == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
        }
        a.gw = circ->d_gwires;
        a.ops = circ->d_ops;
        a.row_of_gate = circ->d_row_of_gate;
        a.in = s->d_io;
        a.out = s->d_io + nin;
        a.ngates = ngates;
        a.first_tmp = first_tmp;
        a.first_out = first_out;
        // (2) byte size of this call's serialisation (depends on in[] / out[]: ids above 0xffff take the long form)
        hipLaunchKernelGGL(k_ser_sizes, dim3(nblocks), dim3(kSerThreads), 0, st, a, s->d_boff);
        hipLaunchKernelGGL(k_ser_scan, dim3(1), dim3(1024), 0, st, s->d_boff, nblocks);
        GC_HIP(hipMemcpyAsync(&s->need_host[slot], s->d_boff + nblocks, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    }
    tr.lap("uploads + sizes");
    // (3) input labels through in[] (Get, :131-141), garble, outputs into the store (Set, :143-157) — all on the device
    gc_batch *b = nullptr;
    int rc = gc_pass_dev(circ, false, s->key.data(), s->key.size(), &s->r, s->store.d, s->d_io, s->d_io + nin + nout, nullptr, 0, &b);
    if (rc != GC_OK) return rc;
    for (uint32_t j = 0; j < nout; j++)
        if (first_out + j >= first_tmp) s->store.on_dev[out[j]] = 1;
    // (4) wire format (:391-446) written by the device at the scanned offsets
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipLaunchKernelGGL(k_ser_write, dim3(nblocks), dim3(kSerThreads), 0, st, a, s->d_boff, b->d_T, b->g.lt, s->d_bytes[slot]);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(s->done[slot], st);
        if (e != hipSuccess) {
            set_error("gc_stream_garble", e);
            rc = GC_E_HIP;
        }
    }
    gc_circ_release_batch(circ, b);  // later passes on the same stream may reuse it: stream order protects the tables
    if (rc == GC_OK) s->pending++;
    tr.lap("enqueue pass + serialiser");
    return rc;
} catch (...) {
    return gc::on_exception();
}

int gc_stream_garble_finish(gc_stream *s, uint8_t *buf, size_t cap, size_t *written) try {
    if (!s || !buf || !written || s->pending == 0) return GC_E_ARG;
    StreamTrace tr;
    const uint32_t slot = s->head;
    s->head ^= 1u;
    s->pending--;
    gc_ctx *ctx = s->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    GC_HIP(hipEventSynchronize(s->done[slot]));
    const uint64_t need = s->need_host[slot];
    *written = (size_t)need;
    if (need > cap) return GC_E_ARG;
    if (need) {  // on its own stream: the next circuit's kernels are already queued on the ctx stream
        GC_HIP(hipMemcpyAsync(buf, s->d_bytes[slot], (size_t)need, hipMemcpyDeviceToHost, s->copy_stream));
        GC_HIP(hipStreamSynchronize(s->copy_stream));
    }
    tr.lap("wait + d2h");
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_stream_garble(gc_stream *s, const gc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                     uint32_t nin, const uint32_t *out, uint32_t nout, uint8_t *buf, size_t cap, size_t *written) {
    if (!s || !buf || !written || s->pending) return GC_E_ARG;
    int rc = gc_stream_garble_begin(s, gates, ngates, nwires, in, nin, out, nout);
    if (rc != GC_OK) return rc;
    return gc_stream_garble_finish(s, buf, cap, written);
}

// ---- streaming evaluator ----------------------------------------------------------------

}  // extern "C"

// Byte skeleton of a block the evaluator has parsed before.  A stream repeats its circuits (every call of a function
// compiles to the same block up to the global wires it is bound to), and the garbler serialises a given circuit the same
// way every time: the same op / flag bytes and tmp ids at the same offsets, only the table rows and the global ids
// differ.  A new block that equals a skeleton on every other byte, and whose global ids repeat in the same pattern, IS
// that circuit: no gate is decoded, the rows are copied out, the global ids are read at their known offsets.
struct EvalSkel {
    struct Chunk {
        uint32_t cmp;    // bytes that must equal the reference block
        uint16_t skip;   // then a global id field (2 / 4 bytes), 0: none
        uint16_t nrows;  // then table rows (16 bytes each)
    };
    size_t nbytes = 0;
    uint32_t nrows = 0, nin = 0, nout = 0;
    gc_circ *circ = nullptr;                 // owned by the stream's circuit cache
    std::vector<uint8_t> bytes;              // the reference block
    std::vector<Chunk> chunks;
    std::vector<uint32_t> gf_off;            // global id fields in stream order: byte offset | 1 << 31 for 4-byte ids
    std::vector<uint32_t> gf_canon;          // index of the first field that names the same wire (the repeat pattern)
    std::vector<uint32_t> in_gf, out_gf;     // field of input k (its first read) / of the k-th global write
    std::vector<uint8_t> out_live;           // 0: a later gate of the block writes the same wire (streaming.Set: last wins)
};

struct gc_stream_eval {
    gc_ctx *ctx = nullptr;
    std::vector<uint8_t> key;
    DevStore store;  // StreamEval.wires (global store), device-resident
    CircCache cache;
    std::vector<uint32_t> io_host;  // indices of this block's inputs, then of its global outputs (0xffffffff: superseded)
    uint32_t *d_io = nullptr;
    size_t io_cap = 0;
    // per-circuit scratch, kept across calls: last writer of every tmp / global wire with a generation stamp
    std::vector<uint64_t> last_t, last_w;  // per tmp / global wire: generation stamp << 32 | current id (one load per look-up)
    uint32_t gen = 0;
    std::vector<gc_gate> gates;       // only materialised for a circuit the cache does not know
    std::vector<CircKey> keys;        // the block's gates as packed records {in0, in1, out, op}
    std::vector<uint64_t> dst_pack;   // per gate: destination index | tmp flag << 32 (parser scratch, kept across calls)
    std::vector<uint32_t> in_idx, id_of;
    std::unordered_map<uint64_t, std::vector<EvalSkel>> skels;  // by (ngates, ntmp); a few byte layouts per key
    std::vector<uint32_t> gf_ids, wr_ids;   // scratch: the block's global ids by field / the wires it writes
    EvalSkel rec;                           // skeleton of the block being parsed (kept when the parse succeeds)
    bool use_skels = true;                  // GC_STREAM_NO_SKELETON (read at creation): every block is parsed
    uint64_t n_parsed = 0, n_matched = 0;
    size_t skel_bytes = 0;                  // reference bytes held by skels (capped: the blocks are the peer's data)
    // table rows of the block being parsed, in pinned memory (true asynchronous H2D); two buffers: the copy of block k
    // may still be in flight while block k + 1 is parsed
    gc_label *slab_pin[2] = {nullptr, nullptr};
    size_t slab_cap[2] = {0, 0};
    hipEvent_t slab_ev[2] = {nullptr, nullptr};
    uint32_t slab_turn = 0;
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
        e->use_skels = std::getenv("GC_STREAM_NO_SKELETON") == nullptr;
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
    if (e->ctx) {
        (void)hipSetDevice(e->ctx->device);
        (void)hipStreamSynchronize(e->ctx->stream);
    }
    for (auto &kv : e->cache) gc_circ_free(kv.second.circ);
    if (e->d_io) (void)hipFree(e->d_io);
    for (int i = 0; i < 2; i++) {
        if (e->slab_pin[i]) (void)hipHostFree(e->slab_pin[i]);
        if (e->slab_ev[i]) (void)hipEventDestroy(e->slab_ev[i]);
    }
    e->store.release();
    delete e;
}

int gc_stream_eval_set_wire(gc_stream_eval *e, uint32_t w, const gc_label *l) try {
    if (!e || !l) return GC_E_ARG;
    e->store.set(w, *l);
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_stream_eval_stats(const gc_stream_eval *e, uint64_t *parsed, uint64_t *matched) {
    if (!e) return GC_E_ARG;
    if (parsed) *parsed = e->n_parsed;
    if (matched) *matched = e->n_matched;
    return GC_OK;
}

int gc_stream_eval_get_wire(gc_stream_eval *e, uint32_t w, gc_label *l) {
    if (!e || !l) return GC_E_ARG;
    return e->store.get(e->ctx, w, l);
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
    e->store.ensure(nwires);  // InitCircuit(numWires, numTmpWires)
    if (ngates == 0) return GC_OK;
    StreamTrace tr;
    // Parse the gate stream (stream_evaluator.go:272-345) into an SSA gate list.  Wire ids of the device circuit:
    //   [0, nin)            wires read before this circuit writes them, in order of first use
    //   then one id per gate: first the gates that write a tmp wire, last the gates that write a global wire —
    //   those are the circuit's outputs and the only labels that come back (streaming.Set, :346-432).
    // "Who wrote (tmp, idx) last" is an array look-up with a generation stamp (no clearing between circuits).
    // A tmp wire is private to its OpCircuit block (stream_garble.go:131-157 gives every non-input, non-output
    // wire of the circuit a tmp id, written by a gate before any gate reads it): a block that reads a tmp it has
    // not written is rejected instead of evaluated on a stale label.
    if (e->last_t.size() < ntmp) {
        e->last_t.resize(ntmp, 0);
    }
    std::vector<CircKey> &gates = e->keys;
    const uint32_t sb = e->slab_turn & 1u;
    e->slab_turn++;
    {
        GC_HIP(hipSetDevice(e->ctx->device));
        if (!e->slab_ev[sb]) GC_HIP(hipEventCreateWithFlags(&e->slab_ev[sb], hipEventDisableTiming));
        else GC_HIP(hipEventSynchronize(e->slab_ev[sb]));  // the H2D of the block two calls ago (long done)
        const size_t want = (size_t)ngates * 3 + 1;
        if (e->slab_cap[sb] < want) {
            if (e->slab_pin[sb]) (void)hipHostFree(e->slab_pin[sb]);
            e->slab_pin[sb] = nullptr;
            e->slab_cap[sb] = 0;
            GC_HIP(hipHostMalloc((void **)&e->slab_pin[sb], (want + want / 2) * sizeof(gc_label), hipHostMallocDefault));
            e->slab_cap[sb] = want + want / 2;
        }
    }
    gc_label *slab = e->slab_pin[sb];
    size_t nrows = 0;
    tr.lap("eval: row buffer free");
    auto load_be64 = [](const uint8_t *p) {
        uint64_t v;
        std::memcpy(&v, p, 8);
        return __builtin_bswap64(v);
    };
    auto field_id = [&](uint32_t f) -> uint32_t {  // global id field (offset | 1 << 31 for the 4-byte form)
        const uint8_t *q = buf + (f & 0x7fffffffu);
        return (f >> 31) ? ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]
                         : ((uint32_t)q[0] << 8) | q[1];
    };
    // repeat pattern of a block's global ids: per field, the index of the first field naming the same wire
    // (stamps in last_w under a generation of their own)
    auto canon_of = [&](const std::vector<uint32_t> &gf_off, std::vector<uint32_t> *ids, const uint32_t *expect,
                        std::vector<uint32_t> *canon) -> bool {
        if (++e->gen == 0) {
            std::fill(e->last_t.begin(), e->last_t.end(), 0);
            std::fill(e->last_w.begin(), e->last_w.end(), 0);
            e->gen = 1;
        }
        const uint64_t stamp = (uint64_t)e->gen << 32;
        ids->resize(gf_off.size());
        if (canon) canon->resize(gf_off.size());
        for (uint32_t f = 0; f < gf_off.size(); f++) {
            const uint32_t id = field_id(gf_off[f]);
            if (id >= nwires) return false;
            if (id >= e->last_w.size()) e->last_w.resize((size_t)id + 1 + e->last_w.size() / 2, 0);
            uint32_t first = f;
            if ((e->last_w[id] >> 32) == e->gen) first = (uint32_t)e->last_w[id];
            else e->last_w[id] = stamp | f;
            if (expect && expect[f] != first) return false;
            if (canon) (*canon)[f] = first;
            (*ids)[f] = id;
        }
        return true;
    };
    gc_circ *circ = nullptr;
    uint32_t nin = 0, nout = 0;
    size_t pos = 0;
    std::vector<uint32_t> &gf_ids = e->gf_ids, &wr_ids = e->wr_ids;
    // ---- a block seen before, up to its rows and global ids?
    {
        auto it = e->skels.find(((uint64_t)ngates << 32) | ntmp);
        if (e->use_skels && it != e->skels.end())
            for (const EvalSkel &sk : it->second) {
                if (sk.nbytes > len) continue;
                const uint8_t *p = buf, *q = sk.bytes.data();
                size_t nr = 0;
                bool same = true;
                for (const EvalSkel::Chunk &c : sk.chunks) {
                    uint64_t acc = 0;
                    uint32_t i = 0;
                    for (; i + 8 <= c.cmp; i += 8) {
                        uint64_t x, y;
                        std::memcpy(&x, p + i, 8);
                        std::memcpy(&y, q + i, 8);
                        acc |= x ^ y;
                    }
                    if (i < c.cmp) {
                        if (c.cmp >= 8) {  // the last, partial word: re-read the run's final 8 bytes
                            uint64_t x, y;
                            std::memcpy(&x, p + c.cmp - 8, 8);
                            std::memcpy(&y, q + c.cmp - 8, 8);
                            acc |= x ^ y;
                        } else {
                            for (; i < c.cmp; i++) acc |= (uint64_t)(p[i] ^ q[i]);
                        }
                    }
                    if (acc) {
                        same = false;
                        break;
                    }
                    p += c.cmp + c.skip;
                    q += c.cmp + c.skip;
                    for (uint32_t r = 0; r < c.nrows; r++, p += 16) slab[nr++] = gc_label{load_be64(p), load_be64(p + 8)};
                    q += 16u * c.nrows;
                }
                if (!same || !canon_of(sk.gf_off, &gf_ids, sk.gf_canon.data(), nullptr)) continue;
                circ = sk.circ;
                nin = sk.nin, nout = sk.nout;
                nrows = nr;
                pos = sk.nbytes;
                e->io_host.resize((size_t)nin + nout + 1);
                for (uint32_t k = 0; k < nin; k++) e->io_host[k] = gf_ids[sk.in_gf[k]];
                wr_ids.resize(nout);
                for (uint32_t k = 0; k < nout; k++) {
                    wr_ids[k] = gf_ids[sk.out_gf[k]];
                    e->io_host[nin + k] = sk.out_live[k] ? wr_ids[k] : 0xffffffffu;
                }
                break;
            }
    }
    if (circ) {
        e->n_matched++;
        tr.lap("eval: skeleton match");
    }
    if (!circ) {
    if (++e->gen == 0) {  // stamp wrap-around
        std::fill(e->last_t.begin(), e->last_t.end(), 0);
        std::fill(e->last_w.begin(), e->last_w.end(), 0);
        e->gen = 1;
    }
    const uint32_t gen = e->gen;
    gates.resize(ngates);
    // The cache is keyed on the block AS PARSED — {in0, in1, writes-a-tmp, op} per gate, operands named by the gate
    // that wrote them (bit 31: the k-th distinct input) — which fixes the device circuit completely; its wire ids
    // (inputs, tmp-writing gates, global-writing gates) are only worked out when the cache does not know the block.
    std::vector<uint64_t> &glob = e->dst_pack;  // (gate << 32 | global wire) of the gates that write a global wire
    std::vector<uint32_t> &inputs = e->in_idx;  // no per-call allocation: a block has ~10^5 gates
    glob.clear();
    inputs.clear();
    EvalSkel &rec = e->rec;  // the block's skeleton, recorded on the way
    rec.chunks.clear(), rec.gf_off.clear(), rec.in_gf.clear(), rec.out_gf.clear();
    size_t run_start = 0;
    bool rec_ok = len < 0x7fffffffu;
    auto cut = [&](size_t at, uint32_t skip, uint32_t rows) {  // the compare run ends at `at`: a global id or rows follow
        rec.chunks.push_back(EvalSkel::Chunk{(uint32_t)(at - run_start), (uint16_t)skip, (uint16_t)rows});
        run_start = at + skip + 16u * (size_t)rows;
    };
    CircuitHash ph(ngates, 0, 0, 0);
    static const uint8_t kRowsOf[5] = {0, 0, 2, 3, 1}, kWiresOf[5] = {3, 3, 3, 3, 2};
    for (uint32_t g = 0; g < ngates; g++) {
        if (pos + 1 > len) return GC_E_ROWS;
        uint8_t gop = buf[pos++];
        const bool at = gop & 0x80, bt = gop & 0x40, ct = gop & 0x20, shortf = gop & 0x10;
        gop &= 0x0f;
        if (gop > GC_INV) return GC_E_GATE;  // "invalid operation"
        const int nw = kWiresOf[gop];
        const uint32_t rows = kRowsOf[gop];
        uint32_t w[3] = {0, 0, 0};
        const size_t idpos = pos;
        const uint32_t idsz = shortf ? 2u : 4u;
        if (shortf) {
            if (pos + 2 * (size_t)nw + 16 * (size_t)rows > len) return GC_E_ROWS;
            for (int i = 0; i < nw; i++) w[i] = ((uint32_t)buf[pos + 2 * i] << 8) | buf[pos + 2 * i + 1];
            pos += 2 * (size_t)nw;
        } else {
            if (pos + 4 * (size_t)nw + 16 * (size_t)rows > len) return GC_E_ROWS;
            for (int i = 0; i < nw; i++) {
                uint32_t v;
                std::memcpy(&v, buf + pos + 4 * i, 4);
                w[i] = __builtin_bswap32(v);
            }
            pos += 4 * (size_t)nw;
        }
        auto global_field = [&](int i) -> uint32_t {  // field i of this gate names a global wire: its field number
            const size_t o = idpos + (size_t)idsz * i;
            cut(o, idsz, 0);
            rec.gf_off.push_back((uint32_t)o | (shortf ? 0u : 0x80000000u));
            return (uint32_t)rec.gf_off.size() - 1;
        };
        int err = GC_OK;
        auto use = [&](bool t, uint32_t idx, int field) -> uint32_t {  // current id of a wire; bit 31: a circuit input
            if (t) {
                if (idx >= ntmp || (e->last_t[idx] >> 32) != gen) {
                    err = GC_E_ARG;
                    return 0;
                }
                return (uint32_t)e->last_t[idx];
            }
            if (idx >= nwires) {
                err = GC_E_ARG;
                return 0;
            }
            const uint32_t f = global_field(field);
            if (idx >= e->last_w.size()) e->last_w.resize((size_t)idx + 1 + e->last_w.size() / 2, 0);
            if ((e->last_w[idx] >> 32) == gen) return (uint32_t)e->last_w[idx];
            const uint32_t id = 0x80000000u | (uint32_t)inputs.size();
            inputs.push_back(idx);
            rec.in_gf.push_back(f);
            e->last_w[idx] = ((uint64_t)gen << 32) | id;
            return id;
        };
        CircKey &k = gates[g];
        k.in0 = use(at, w[0], 0);
        k.in1 = nw == 3 ? use(bt, w[1], 1) : k.in0;
        if (err != GC_OK) return err;
        k.op = gop;
        k.out = ct ? 1u : 0u;
        ph.mix(g, k);
        const uint32_t ci = w[nw - 1];
        if (ct) {
            if (ci >= ntmp) return GC_E_ARG;
            e->last_t[ci] = ((uint64_t)gen << 32) | g;
        } else {
            if (ci >= nwires) return GC_E_ARG;
            rec.out_gf.push_back(global_field(nw - 1));
            if (ci >= e->last_w.size()) e->last_w.resize((size_t)ci + 1 + e->last_w.size() / 2, 0);
            e->last_w[ci] = ((uint64_t)gen << 32) | g;
            glob.push_back(((uint64_t)g << 32) | ci);
        }
        if (rows) cut(pos, 0, rows);
        for (uint32_t r = 0; r < rows; r++) {
            slab[nrows++] = gc_label{load_be64(buf + pos), load_be64(buf + pos + 8)};
            pos += 16;
        }
    }
    cut(pos, 0, 0);
    tr.lap("eval: parse");
    nin = (uint32_t)inputs.size(), nout = (uint32_t)glob.size();
    const uint32_t n_tmp = ngates - nout;
    const uint32_t cw = nin + ngates;
    const uint64_t h = ((ph.done() ^ nin) * CircuitHash::kPrime ^ nout) * CircuitHash::kPrime;
    // device circuit, cached by content
    circ = cache_find_keys(e->cache, h, gates, cw, nin, nout);
    if (!circ) {
        // wire ids of the device circuit: inputs, then the tmp-writing gates, then the global-writing gates
        int st = GC_OK;
        std::vector<uint32_t> &id_of = e->id_of;
        id_of.resize(ngates);
        uint32_t kt = 0, kg = 0;
        for (uint32_t g = 0; g < ngates; g++) id_of[g] = gates[g].out ? nin + kt++ : nin + n_tmp + kg++;
        std::vector<gc_gate> &full = e->gates;
        full.assign(ngates, gc_gate{});
        auto fix = [&](uint32_t v) { return (v & 0x80000000u) ? (v & 0x7fffffffu) : id_of[v]; };
        for (uint32_t g = 0; g < ngates; g++) {
            full[g].in0 = fix(gates[g].in0);
            full[g].in1 = gates[g].op == GC_INV ? 0 : fix(gates[g].in1);
            full[g].out = id_of[g];
            full[g].op = (uint8_t)gates[g].op;
        }
        circ = gc_circ_load(e->ctx, full.data(), ngates, cw, nin, nout, &st);
        if (!circ) return st;
        cache_put_keys(e->cache, h, circ, gates, cw, nin, nout);
    }
    tr.lap("eval: hash + cache");
    // Input labels are gathered from, output labels scattered into, the device-resident store: nothing waits for the
    // GPU, so the parsing of the next block overlaps the evaluation of this one.  Only the LAST gate of the block that
    // writes a global wire stores it (streaming.Set in gate order, :346-432: the last write wins).
    e->io_host.resize((size_t)nin + nout + 1);
    for (uint32_t i = 0; i < nin; i++) e->io_host[i] = inputs[i];
    wr_ids.resize(nout);
    rec.out_live.resize(nout);
    for (uint32_t k = 0; k < nout; k++) {
        const uint32_t g = (uint32_t)(glob[k] >> 32), idx = (uint32_t)glob[k];
        wr_ids[k] = idx;
        rec.out_live[k] = (uint32_t)e->last_w[idx] == g;
        e->io_host[nin + k] = rec.out_live[k] ? idx : 0xffffffffu;
    }
    // keep the skeleton: the next block of this circuit is matched byte-wise instead of parsed
    e->n_parsed++;
    constexpr size_t kSkelCap = (size_t)1 << 30;  // beyond 1 GiB of reference blocks every new circuit is parsed each time
    if (e->use_skels && rec_ok && e->skel_bytes + pos <= kSkelCap) {
        std::vector<EvalSkel> &v = e->skels[((uint64_t)ngates << 32) | ntmp];
        if (v.size() >= 8) {
            e->skel_bytes -= v.front().bytes.size();
            v.erase(v.begin());
        }
        EvalSkel sk;
        sk.nbytes = pos;
        sk.nrows = (uint32_t)nrows, sk.nin = nin, sk.nout = nout;
        sk.circ = circ;
        sk.bytes.assign(buf, buf + pos);
        sk.chunks = rec.chunks;
        sk.gf_off = rec.gf_off;
        sk.in_gf = rec.in_gf, sk.out_gf = rec.out_gf, sk.out_live = rec.out_live;
        if (canon_of(sk.gf_off, &gf_ids, nullptr, &sk.gf_canon)) {
            e->skel_bytes += sk.bytes.size();
            v.push_back(std::move(sk));
        }
    }
    }  // parsed
    gc_ctx *ctx = e->ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        GC_HIP(hipSetDevice(ctx->device));
        int rcs = e->store.flush(ctx);
        if (rcs != GC_OK) return rcs;
        hipError_t er = grow(&e->d_io, &e->io_cap, e->io_host.size());
        if (er == hipSuccess)
            er = hipMemcpyAsync(e->d_io, e->io_host.data(), e->io_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
        if (er != hipSuccess) {
            set_error("gc_stream_eval_circuit", er);
            return er == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
        }
    }
    gc_batch *b = nullptr;
    int rc = gc_pass_dev(circ, true, e->key.data(), e->key.size(), nullptr, e->store.d, e->d_io, e->d_io + nin, slab, nrows, &b);
    if (rc != GC_OK) return rc;
    GC_HIP(hipEventRecord(e->slab_ev[sb], ctx->stream));
    gc_circ_release_batch(circ, b);
    for (uint32_t k = 0; k < nout; k++) e->store.on_dev[e->wr_ids[k]] = 1;
    tr.lap("eval: enqueue");
    *consumed = pos;
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

}  // extern "C"
