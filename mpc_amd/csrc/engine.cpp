// engine.cpp — C ABI of libgcengine.so: contexts, circuits, batches, host-buffer entry points.
// (HIP host code; kernels are in gc_kernels.hip / ot_kernels.hip.)
#include "engine.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>

namespace gc {

thread_local char tls_error[512] = "";

void set_error(const char *what, hipError_t e) {
    std::snprintf(tls_error, sizeof tls_error, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    // a failed runtime call also leaves the thread's "last error" set: clear it, or the next launch's
    // hipGetLastError() reports this failure again
    (void)hipGetLastError();
}

int on_exception() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        std::snprintf(tls_error, sizeof tls_error, "host allocation failed (std::bad_alloc)");
        return GC_E_NOMEM;
    } catch (const std::length_error &e) {
        std::snprintf(tls_error, sizeof tls_error, "host allocation failed (%s)", e.what());
        return GC_E_NOMEM;
    } catch (const std::exception &e) {
        std::snprintf(tls_error, sizeof tls_error, "internal error: %s", e.what());
        return GC_E_HIP;
    } catch (...) {
        std::snprintf(tls_error, sizeof tls_error, "internal error (unknown exception)");
        return GC_E_HIP;
    }
}

}  // namespace gc

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues per device (4 by default), and two
// streams that share a queue run one after the other.  A garbler / evaluator stream of this engine uses up to seven HIP streams
// that are meant to run side by side (ctx stream, copy / serialiser / upload streams, the deep lanes of stream_lanes.cpp): a host
// that streams should start with GPU_MAX_HW_QUEUES=8 in its environment (the runtime reads it on the first HIP call of the
// process).  The library does NOT set it: changing the environment of a host process from a constructor races
// with its other threads and changes the runtime for everybody in it.  Without it the engine still works: lanes that
// share the ctx stream's queue are detected and not used (DeepLanes::setup_ctx).  mpc_amd/engine.py, bench.py's children and
// tools/stream_driver.c set it for themselves, before their first HIP call.

using namespace gc;

namespace {
// developer aid: GC_TRACE=1 prints the wall-clock laps of the host-buffer calls to stderr
struct Trace {
    bool on;
    const char *name;
    std::chrono::steady_clock::time_point t0, last;
    explicit Trace(const char *n) : on(std::getenv("GC_TRACE") != nullptr), name(n) {
        if (on) t0 = last = std::chrono::steady_clock::now();
    }
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gc trace] %s: %-18s %8.3f ms\n", name, what,
                     std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
    ~Trace() {
        if (on)
            std::fprintf(stderr, "[gc trace] %s: total              %8.3f ms\n", name,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};
}  // namespace

// Contexts alive in the process.  The engine records a pass as a hipGraph on its own only while there is ONE: a kernel launch
// on another thread's stream invalidates a capture under way here (hipErrorStreamCaptureInvalidated from the launches inside
// it, measured with three streams on three threads and GC_NO_COOP) — with several contexts the same kernels are launched
// directly.  (gc_ctx_capture_begin / _end, the caller's own recording, is the caller's to keep single-threaded.)
static std::atomic<int> &live_contexts() {
    static std::atomic<int> n{0};
    return n;
}

// ---- cooperative one-launch passes of ONE instance (kernels.h: launch_coop) ------------------------------------------------
// First use: allocate the barrier state and run the self-test (are the kCoopGroups workgroups on one XCD, do values stored
// before the barrier arrive behind it); GC_NO_COOP keeps the level launches.  Called with ctx->mu held or from one thread.
static bool coop_ready(gc_ctx *c) {
    if (c->capturing) return false;  // (recorded pipelines keep the level launches)
    if (c->coop_state != 0) return c->coop_state > 0;
    c->coop_state = -1;
    if (std::getenv("GC_NO_COOP")) return false;
    if (hipMalloc((void **)&c->d_coop, sizeof(CoopCtl)) != hipSuccess) return false;
    // the ctx's error word (gc_ctx_err_word: kernels of step groups in flight may raise it — it is never written from here) and,
    // apart from it, a read-back buffer of its own for the head of the control block after the self-test
    if (!gc_ctx_err_word(c)) return false;
    uint32_t head[4];
    std::memset(head, 0xff, sizeof head);
    launch_coop_selftest(c->d_coop, c->stream);
    if (hipMemcpyAsync(head, c->d_coop, sizeof head, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
    const CoopCtl *h = (const CoopCtl *)head;  // count, error, bad, ticks: the first four words
    const bool ok = h->error == 0 && h->bad == 0 && h->count == 128u * kCoopGroups;
    if (std::getenv("GC_TRACE"))
        std::fprintf(stderr, "[gc trace] coop self-test: %s, error %u bad %u, 128 barriers in %u ticks\n", ok ? "passed" : "FAILED",
                     h->error, h->bad, h->ticks);
    // the passes find the counters at zero and leave them so
    CoopCtl zero{};
    if (const char *f = std::getenv("GC_COOP_FORCE_TIMEOUT")) zero.drop_at = (uint32_t)std::max(0, std::atoi(f));  // (testing)
    if (hipHostGetDevicePointer((void **)&zero.host_err, c->h_coop_err, 0) != hipSuccess || !zero.host_err) return false;
    if (hipMemcpy(c->d_coop, &zero, sizeof zero, hipMemcpyHostToDevice) != hipSuccess) return false;
    if (ok) c->coop_state = 1;
    return ok;
}

// The pinned word of the ctx that kernels raise when a bounded wait on the device runs out (coop_ready allocates it too):
// its device address, nullptr if there is none to be had.  Called with ctx->mu held.
uint32_t *gc_ctx_err_word(gc_ctx *c) {
    if (!c->h_coop_err) {
        if (hipHostMalloc((void **)&c->h_coop_err, 256, hipHostMallocPortable) != hipSuccess) {
            c->h_coop_err = nullptr;
            (void)hipGetLastError();
            return nullptr;
        }
        std::memset(c->h_coop_err, 0, 256);
    }
    uint32_t *d = nullptr;
    if (hipHostGetDevicePointer((void **)&d, c->h_coop_err, 0) != hipSuccess) return nullptr;
    return d;
}

namespace gc {
constexpr size_t kBufMin = (size_t)64 << 10;
// what a ctx's lists may hold (idle buffers; GC_CTX_CACHE_MIB = "<device MiB>,<pinned MiB>" overrides it)
static void cache_caps(size_t *dev, size_t *pin) {
    static size_t d = 0, p = 0;
    static std::once_flag once;
    std::call_once(once, [] {
        d = (size_t)4 << 30, p = (size_t)1 << 30;
        if (const char *v = std::getenv("GC_CTX_CACHE_MIB")) {
            unsigned long long a = 0, b = 0;
            const int n = std::sscanf(v, "%llu,%llu", &a, &b);
            if (n >= 1) d = (size_t)a << 20;
            if (n >= 2) p = (size_t)b << 20;
        }
    });
    *dev = d, *pin = p;
}
// every idle buffer of the kind back to the runtime (an allocation has failed: memory parked here — outgrown size classes,
// the lists of a stream long gone — is the first thing to give back)
static void ctx_buf_trim(gc_ctx *c, bool pinned) {
    std::vector<gc_ctx::CachedBuf> drop;
    {
        std::lock_guard<std::mutex> lk(c->cache_mu);
        drop.swap(pinned ? c->pin_cache : c->dev_cache);
        (pinned ? c->pin_cached : c->dev_cached) = 0;
    }
    for (auto &b : drop) {
        if (pinned) (void)hipHostFree(b.p);
        else (void)hipFree(b.p);
    }
}
hipError_t ctx_buf_get(gc_ctx *c, bool pinned, size_t need, void **p, size_t *cap) {
    size_t want = kBufMin;
    while (want < need) want *= 2;
    {
        std::lock_guard<std::mutex> lk(c->cache_mu);
        auto &list = pinned ? c->pin_cache : c->dev_cache;
        for (size_t i = list.size(); i-- > 0;)
            if (list[i].cap == want) {
                *p = list[i].p;
                *cap = want;
                (pinned ? c->pin_cached : c->dev_cached) -= want;
                list.erase(list.begin() + (long)i);
                return hipSuccess;
            }
    }
    void *q = nullptr;
    hipError_t e = pinned ? hipHostMalloc(&q, want, hipHostMallocDefault) : hipMalloc(&q, want);
    if (e != hipSuccess) {  // no memory while this ctx sits on idle buffers of other sizes: give them back and ask again
        (void)hipGetLastError();
        ctx_buf_trim(c, pinned);
        if (!pinned) ctx_buf_trim(c, true);  // (pinned pages count against the host, but a device allocation may need staging too)
        e = pinned ? hipHostMalloc(&q, want, hipHostMallocDefault) : hipMalloc(&q, want);
    }
    if (e != hipSuccess) return e;
    *p = q;
    *cap = want;
    return hipSuccess;
}
void ctx_buf_put(gc_ctx *c, bool pinned, void *p, size_t cap) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(c->cache_mu);
        size_t &held = pinned ? c->pin_cached : c->dev_cached;
        size_t dev_max = 0, pin_max = 0;
        cache_caps(&dev_max, &pin_max);
        if (held + cap <= (pinned ? pin_max : dev_max)) {
            (pinned ? c->pin_cache : c->dev_cache).push_back(gc_ctx::CachedBuf{p, cap});
            held += cap;
            return;
        }
    }
    if (pinned) (void)hipHostFree(p);
    else (void)hipFree(p);
}
}  // namespace gc

// A cooperative pass that lost a workgroup (its bounded wait ran out) is done again on the device by the stand-by workgroup of
// the same launch, before anything that follows it on the stream — and the stand-by is who tells the host: the pinned word
// becomes 1 when a pass HAS BEEN done again (results are good; the host counts it and keeps to the level launches from here on:
// a ctx whose passes do not stay resident together — three or more streams on one GPU — would pay 36 ms per pass), 2 when the
// stand-by could not do so (the pass had not ended within its ~2 s backstop, or host and device disagree on the pass count):
// labels or table rows of that pass are garbage then, and every call that hands results out reports GC_E_HIP.
int gc_ctx_coop_check(gc_ctx *c) {
    if (!c || !c->h_coop_err || *c->h_coop_err == 0) return GC_OK;
    const uint32_t what = *c->h_coop_err;
    c->coop_state = -1;
    if (what >= 2) {  // (the word stays up: the ctx's results cannot be trusted any more)
        if (what == 3)
            std::snprintf(gc::tls_error, sizeof gc::tls_error, "a unit of a step group waited ~2 s for a unit of the same launch and ran without it");
        else
            std::snprintf(gc::tls_error, sizeof gc::tls_error,
                          "a cooperative one-instance pass lost a workgroup and could not be repeated on the device "
                          "(stand-by of pass %u: device has ended %u passes, host launched %u; arrivals %u, leavers %u, error flag %u, "
                          "waited %u Ki ticks, %u levels)",
                          c->h_coop_err[1], c->h_coop_err[2], c->coop_launched, c->h_coop_err[3], c->h_coop_err[4], c->h_coop_err[5],
                          c->h_coop_err[6], c->h_coop_err[7]);
        return GC_E_HIP;
    }
    *c->h_coop_err = 0;
    c->coop_timeouts++;
    if (std::getenv("GC_TRACE"))
        std::fprintf(stderr, "[gc trace] a cooperative one-instance pass lost a workgroup and was repeated on the device; level launches from here on\n");
    return GC_OK;
}

extern "C" {

const char *gc_strerror(int status) {
    switch (status) {
    case GC_OK: return "ok";
    case GC_E_KEYSIZE: return "crypto/aes: invalid key size";
    case GC_E_RAND: return "random stream too short";
    case GC_E_GATE: return "invalid gate type";
    case GC_E_ROWS: return "corrupted circuit: garbled table size";
    case GC_E_ARG: return "invalid argument";
    case GC_E_HIP: return "HIP runtime error";
    case GC_E_NOMEM: return "out of memory";
    case GC_E_WIRE: return "gate input wire not set";
    default: return "unknown status";
    }
}

const char *gc_last_error(void) { return tls_error; }
int gc_abi_version(void) { return GC_ABI_VERSION; }

int gc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- context ---------------------------------------------------------------------------------

gc_ctx *gc_ctx_create(int device, int *status) try {
    int rc = GC_OK;
    gc_ctx *c = new (std::nothrow) gc_ctx;
    if (!c) rc = GC_E_NOMEM;
    else live_contexts()++;
    if (rc == GC_OK) {
        c->device = device;
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&c->d_te0, 256 * sizeof(uint32_t));
        if (e == hipSuccess)
            e = hipMemcpy(c->d_te0, aes_tables().te0, 256 * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("gc_ctx_create", e);
            rc = GC_E_HIP;
        }
    }
    if (rc != GC_OK && c) {
        gc_ctx_destroy(c);
        c = nullptr;
    }
    if (status) *status = rc;
    return c;
} catch (...) {
    const int rc__ = gc::on_exception();
    if (status) *status = rc__;
    return nullptr;
}

void gc_ctx_destroy(gc_ctx *c) {
    if (!c) return;
    live_contexts()--;
    (void)hipSetDevice(c->device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    if (c->d_te0) (void)hipFree(c->d_te0);
    if (c->d_coop) (void)hipFree(c->d_coop);
    if (c->h_coop_err) (void)hipHostFree(c->h_coop_err);
    for (int b = 0; b < 2; b++) {
        if (c->stage[b]) (void)hipFree(c->stage[b]);
        if (c->ev_k[b]) (void)hipEventDestroy(c->ev_k[b]);
        if (c->ev_c[b]) (void)hipEventDestroy(c->ev_c[b]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    gcs_fuse_cache_free(c);
    for (hipStream_t l : c->lanes) (void)hipStreamDestroy(l);
    for (hipStream_t l : c->lanes_aside) (void)hipStreamDestroy(l);
    for (auto &b : c->dev_cache) (void)hipFree(b.p);
    for (auto &b : c->pin_cache) (void)hipHostFree(b.p);
    delete c;
}

// ---- pinned host memory for the host-buffer API --------------------------------------------------------------
// The literal drop-in calls move 238 KB of tables per aes_128 instance across PCIe.  From pageable memory the runtime
// stages every copy through its own bounce buffers (37 / 46 GB/s measured); from pinned memory the DMA engine reads /
// writes the caller's pages directly (~57 GB/s) and gc_garble / gc_eval pipeline it against their transpose kernels.
// The Go shim backs its pooled scratch (Garbled.Wires / the slab, cf. garble.go:195-225) with gc_host_alloc.
void *gc_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void gc_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int gc_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return GC_E_ARG;
    GC_HIP(hipHostRegister(p, bytes, hipHostRegisterPortable));
    return GC_OK;
}

int gc_host_unregister(void *p) {
    if (!p) return GC_E_ARG;
    GC_HIP(hipHostUnregister(p));
    return GC_OK;
}

int gc_host_is_pinned(const void *p) {
    if (!p) return 0;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return a.type == hipMemoryTypeHost ? 1 : 0;
}

// ---- device memory for hosts that own no HIP allocator (the Go shim; bench.py and the tests use the same calls) ------
void *gc_dev_alloc(gc_ctx *c, size_t bytes, int *status) {
    int rc = GC_OK;
    void *p = nullptr;
    if (!c) rc = GC_E_ARG;
    if (rc == GC_OK) {
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) {
            set_error("gc_dev_alloc", e);
            rc = e == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
            p = nullptr;
        }
    }
    if (status) *status = rc;
    return p;
}

void gc_dev_free(gc_ctx *c, void *d) {
    if (!c || !d) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);  // kernels queued on the ctx stream may still read / write it
    (void)hipFree(d);
}

int gc_dev_upload(gc_ctx *c, void *d_dst, const void *src, size_t bytes) {
    if (!c || (bytes && (!d_dst || !src))) return GC_E_ARG;
    if (!bytes) return GC_OK;
    if (c->capturing) return GC_E_ARG;  // a host copy cannot be part of a pipeline graph
    GC_HIP(hipSetDevice(c->device));
    GC_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    // the contract is "src may be reused on return" whatever kind of memory it is (pinned sources are read by the DMA
    // engine after the call returns otherwise)
    GC_HIP(hipStreamSynchronize(c->stream));
    return GC_OK;
}

int gc_dev_download(gc_ctx *c, void *dst, const void *d_src, size_t bytes) {
    if (!c || (bytes && (!dst || !d_src))) return GC_E_ARG;
    if (!bytes) return GC_OK;
    if (c->capturing) return GC_E_ARG;
    GC_HIP(hipSetDevice(c->device));
    GC_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    GC_HIP(hipStreamSynchronize(c->stream));
    return GC_OK;
}

int gc_dev_memset(gc_ctx *c, void *d, int byte_value, size_t bytes) {
    if (!c || (bytes && !d)) return GC_E_ARG;
    if (!bytes) return GC_OK;
    GC_HIP(hipSetDevice(c->device));
    GC_HIP(hipMemsetAsync(d, byte_value, bytes, c->stream));
    return GC_OK;
}

int gc_dev_copy(gc_ctx *c, void *d_dst, const void *d_src, size_t bytes) {
    if (!c || (bytes && (!d_dst || !d_src))) return GC_E_ARG;
    if (!bytes) return GC_OK;
    GC_HIP(hipSetDevice(c->device));
    GC_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return GC_OK;
}

int gc_ctx_sync(gc_ctx *c) {
    if (!c) return GC_E_ARG;
    GC_HIP(hipSetDevice(c->device));
    GC_HIP(hipStreamSynchronize(c->stream));
    return gc_ctx_coop_check(c);
}

void *gc_ctx_stream(gc_ctx *c) { return c ? (void *)c->stream : nullptr; }

int gc_ctx_pci_bus_id(gc_ctx *c, char *buf, size_t len) {
    if (!c || !buf || len < 16) return GC_E_ARG;
    GC_HIP(hipDeviceGetPCIBusId(buf, (int)len, c->device));
    return GC_OK;
}

int gc_ctx_coop_stats(gc_ctx *c, int *state, uint64_t *timeouts) {
    if (!c) return GC_E_ARG;
    (void)gc_ctx_coop_check(c);  // (a pass that has raised the word since the last look)
    if (state) *state = c->coop_state;
    if (timeouts) *timeouts = c->coop_timeouts;
    return GC_OK;
}

// ---- pipeline graphs: record a sequence of device-resident calls once, replay it with one launch ------------
int gc_ctx_capture_begin(gc_ctx *c) {
    if (!c || c->capturing) return GC_E_ARG;
    GC_HIP(hipSetDevice(c->device));
    GC_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    c->capturing = true;
    return GC_OK;
}

int gc_ctx_capture_end(gc_ctx *c, gc_graph **out) try {
    if (!c || !c->capturing || !out) return GC_E_ARG;
    *out = nullptr;
    c->capturing = false;
    hipGraph_t graph = nullptr;
    GC_HIP(hipStreamEndCapture(c->stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        set_error("hipGraphInstantiate", e);
        return GC_E_HIP;
    }
    gc_graph *g = new (std::nothrow) gc_graph;
    if (!g) {
        (void)hipGraphExecDestroy(exec);
        return GC_E_NOMEM;
    }
    g->ctx = c;
    g->exec = exec;
    *out = g;
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_graph_launch(gc_graph *g) {
    if (!g || !g->exec) return GC_E_ARG;
    GC_HIP(hipSetDevice(g->ctx->device));
    GC_HIP(hipGraphLaunch(g->exec, g->ctx->stream));
    return GC_OK;
}

void gc_graph_free(gc_graph *g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    delete g;
}

// ---- circuit ---------------------------------------------------------------------------------

gc_circ *gc_circ_load(gc_ctx *ctx, const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                      uint32_t noutputs, int *status) {
    return gc_circ_load_seg(ctx, gates, ngates, nwires, ninputs, noutputs, nullptr, 0, status);
}

}  // extern "C"

gc_circ *gc_circ_load_seg(gc_ctx *ctx, const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                          uint32_t noutputs, const uint32_t *seg_first, uint32_t nseg, int *status) try {
    int rc = GC_OK;
    gc_circ *c = nullptr;
    if (!ctx) rc = GC_E_ARG;
    if (rc == GC_OK) {
        c = new (std::nothrow) gc_circ;
        if (!c) rc = GC_E_NOMEM;
    }
    if (rc == GC_OK) {
        c->ctx = ctx;
        // the flattened plan (70 % of the build) is deferred: ONE instance of a wide circuit never needs it
        rc = build_plan(gates, ngates, nwires, ninputs, noutputs, &c->plan.p, /*defer_flat=*/true, seg_first, nseg);
    }
    if (rc == GC_OK) {
        const Plan &p = c->plan.p;
        hipError_t e = hipSetDevice(ctx->device);
        auto up = [&](void **dptr, const void *src, size_t bytes) {
            if (e != hipSuccess) return;
            e = hipMalloc(dptr, bytes ? bytes : 16);
            if (e == hipSuccess && bytes) e = hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice);
        };
        up((void **)&c->d_descs, p.descs.data(), p.descs.size() * sizeof(GateDesc));
        up((void **)&c->d_out_slots, p.out_slots.data(), p.out_slots.size() * sizeof(uint32_t));
        up((void **)&c->d_slot_of_wire, p.slot_of_wire.data(), p.slot_of_wire.size() * sizeof(uint32_t));
        up((void **)&c->d_steps, p.levels.data(), p.levels.size() * sizeof(Step));
        {
            std::vector<uint8_t> ops(ngates);
            for (uint32_t i = 0; i < ngates; i++) ops[i] = gates[i].op;
            up((void **)&c->d_ops, ops.data(), ops.size());
            up((void **)&c->d_row_of_gate, p.row_of_gate.data(), p.row_of_gate.size() * sizeof(uint32_t));
        }
        if (e != hipSuccess) {
            set_error("gc_circ_load", e);
            rc = e == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
        }
    }
    if (rc != GC_OK && c) {
        gc_circ_free(c);
        c = nullptr;
    }
    if (status) *status = rc;
    return c;
} catch (...) {
    const int rc__ = gc::on_exception();
    if (status) *status = rc__;
    return nullptr;
}

extern "C" {

void gc_circ_free(gc_circ *c) {
    if (!c) return;
    if (c->flat_thread.joinable()) c->flat_thread.join();
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    for (gc_batch *b : c->pool) gc_batch_free(b);
    c->pool.clear();
    if (c->d_descs) (void)hipFree(c->d_descs);
    if (c->d_out_slots) (void)hipFree(c->d_out_slots);
    if (c->d_slot_of_wire) (void)hipFree(c->d_slot_of_wire);
    if (c->d_steps) (void)hipFree(c->d_steps);
    if (c->d_ops) (void)hipFree(c->d_ops);
    if (c->d_row_of_gate) (void)hipFree(c->d_row_of_gate);
    if (c->d_gwires) (void)hipFree(c->d_gwires);
    if (c->d_fdescs) (void)hipFree(c->d_fdescs);
    if (c->d_fgslot) (void)hipFree(c->d_fgslot);
    if (c->d_fsteps) (void)hipFree(c->d_fsteps);
    if (c->d_in_lds) (void)hipFree(c->d_in_lds);
    if (c->d_fchunks) (void)hipFree(c->d_fchunks);
    if (c->d_fl_prog) (void)hipFree(c->d_fl_prog);
    if (c->d_fl_units) (void)hipFree(c->d_fl_units);
    if (c->d_fl_hgslot) (void)hipFree(c->d_fl_hgslot);
    if (c->d_fl_ogslot) (void)hipFree(c->d_fl_ogslot);
    if (c->d_fl_in_lds) (void)hipFree(c->d_fl_in_lds);
    delete c;
}

// Flattened plan + its device arrays, on first demand (thread-safe; a failure leaves the circuit without a flattened
// plan: the level-walking / HBM-wire kernels serve it)
static void circ_ensure_flat(gc_circ *c) {
    std::lock_guard<std::mutex> lk(c->flat_mu);
    if (c->flat_ready.load(std::memory_order_acquire)) return;
    Plan &p = c->plan.p;
    // flat_ready flips only when the plan is in its final state — complete, or marked "no LDS / flattened plan" —
    // so a later call never launches the LDS kernels on a half-built plan
    struct Done {
        gc_circ *c;
        ~Done() { c->flat_ready.store(true, std::memory_order_release); }
    } done{c};
    auto no_plans = [&] {
        p.n_lds_slots = 0xffffffffu;
        p.info.n_lds_slots = 0xffffffffu;
        p.n_flat_slots = 0xffffffffu;
        p.info.n_flat_slots = 0xffffffffu;
    };
    try {
        finish_flat(&p);
        hipError_t e = hipSetDevice(c->ctx->device);
        auto up = [&](void **dptr, const void *src, size_t bytes) {
            if (e != hipSuccess) return;
            e = hipMalloc(dptr, bytes ? bytes : 16);
            if (e == hipSuccess && bytes) e = hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice);
        };
        // the level-walking fused schedule (fused_lds_kernels.hip: store_all / schedule 2 / circuits without a flattened plan)
        up((void **)&c->d_fdescs, p.fdescs.data(), p.fdescs.size() * sizeof(FDesc));
        up((void **)&c->d_fgslot, p.fgslot.data(), p.fgslot.size() * sizeof(uint32_t));
        up((void **)&c->d_fsteps, p.fsteps.data(), p.fsteps.size() * sizeof(Step));
        up((void **)&c->d_in_lds, p.in_lds.data(), p.in_lds.size() * sizeof(uint16_t));
        up((void **)&c->d_fchunks, p.fchunks.data(), p.fchunks.size() * sizeof(Chunk));
        if (e != hipSuccess) {  // without them only the HBM-wire kernels can serve the circuit
            set_error("circ_ensure_flat", e);
            no_plans();
            return;
        }
        if (p.n_flat_slots == 0xffffffffu) return;
        {
            // the kernels fetch headers / images two units ahead without bounds checks: two zero records and one
            // stage buffer of zero padding behind the real data
            std::vector<uint32_t> prog(p.fl_prog);
            prog.resize(prog.size() + 4 * 1024, 0);
            std::vector<FUnit> units(p.fl_units);
            units.resize(units.size() + 2, FUnit{});
            up((void **)&c->d_fl_prog, prog.data(), prog.size() * sizeof(uint32_t));
            up((void **)&c->d_fl_units, units.data(), units.size() * sizeof(FUnit));
        }
        up((void **)&c->d_fl_hgslot, p.fl_hgslot.data(), p.fl_hgslot.size() * sizeof(uint32_t));
        up((void **)&c->d_fl_ogslot, p.fl_ogslot.data(), p.fl_ogslot.size() * sizeof(uint32_t));
        up((void **)&c->d_fl_in_lds, p.fl_in_lds.data(), p.fl_in_lds.size() * sizeof(uint16_t));
        if (e != hipSuccess) {  // no device arrays: behave as if the circuit had no flattened plan
            set_error("circ_ensure_flat", e);
            p.n_flat_slots = 0xffffffffu;
            p.info.n_flat_slots = 0xffffffffu;
        }
    } catch (...) {  // bad_alloc on a huge streamed circuit: the HBM-wire kernels serve it
        (void)gc::on_exception();
        no_plans();
    }
}

// the flattened plan is being built by the circuit's thread: a pass that starts now keeps its wires in HBM
static bool flat_pending(const gc_circ *c) {
    return c->flat_async.load(std::memory_order_acquire) == 1 && !c->flat_ready.load(std::memory_order_acquire);
}

// introspection wants the whole plan
const gc_plan *gc_circ_plan(const gc_circ *c) {
    if (!c) return nullptr;
    circ_ensure_flat(const_cast<gc_circ *>(c));
    return &c->plan;
}

int gc_circ_set_schedule(gc_circ *c, int schedule) {
    if (!c || schedule < 0 || schedule > 2) return GC_E_ARG;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    c->schedule = schedule ? 1 : 0;
    c->single_phase = schedule == 2;
    return GC_OK;
}

// ---- batch -----------------------------------------------------------------------------------

static void drop_graphs(gc_batch *b);

static void free_buffers(gc_batch *b) {
    if (b->d_W) (void)hipFree(b->d_W);
    if (b->d_T) (void)hipFree(b->d_T);
    if (b->d_R) (void)hipFree(b->d_R);
    b->d_W = b->d_T = b->d_R = nullptr;
    if (b->d_prof) (void)hipFree(b->d_prof);
    b->d_prof = nullptr;
}

// (re)allocate the label / table / R arrays for the batch's schedule (= memory layout)
// live labels the kernel of this batch keeps in LDS: the flattened plan normally, the level-walking plan when
// every wire has to be materialised (store_all) or schedule 2 was asked for
// The flattened kernels are the choice unless every wire has to be materialised (store_all), schedule 2 was asked
// for, or the circuit has no flattened plan.  They address a tile's table rows with 32-bit BYTE offsets:
// slab_rows * 64 instances * 16 B < 2^32 (a circuit with >= 4 Mi table rows does not fit an LDS plan anyway).
// ONE instance of a wide circuit: one launch per level with the gates on the lanes (run_levels); needs no LDS plan
static bool one_wide(const gc_batch *b) {
    return b->schedule == 1 && b->g.batch == 1 && !b->d_prof && wide_for_one_instance(b->circ->plan.p, false);
}

static bool want_flat(const gc_batch *b) {
    if (one_wide(b) || flat_pending(b->circ)) return false;
    circ_ensure_flat(b->circ);
    const Plan &p = b->circ->plan.p;
    return !b->store_all && !b->single_phase && p.n_flat_slots != 0xffffffffu && p.info.slab_rows < (1u << 22);
}

static BatchGeom geom_for(const gc_batch *b) {
    const Plan &p = b->circ->plan.p;
    // (no plan with the wires in LDS yet — the circuit's own thread is building it: the HBM-wire kernels; none of the plan's
    // flattened / LDS fields is read while that thread writes them: constants instead)
    const bool pend = flat_pending(b->circ);
    if (pend) return make_geom(b->g.batch, b->schedule, p.info.nslots, p.info.slab_rows, 0xffffffffu, 6u, false, 0);
    const bool flat = want_flat(b);
    const uint32_t nls = one_wide(b) ? 0xffffffffu : flat ? p.n_flat_slots : p.n_lds_slots;
    // an XOR list spread over 2 / 4 lanes (TI apart) is joined with DPP row shifts: parts * TI <= 16
    const uint32_t max_t = flat ? (p.fl_max_parts >= 4 ? 2u : p.fl_max_parts == 2 ? 3u : 6u) : 6u;
    return make_geom(b->g.batch, b->schedule, p.info.nslots, p.info.slab_rows, nls, max_t, flat, p.fl_unit_stride);
}

static hipError_t alloc_buffers(gc_batch *b) {
    const Plan &p = b->circ->plan.p;
    b->g = geom_for(b);
    const size_t wbytes = (size_t)p.info.nslots * b->g.bstride * sizeof(uint4);
    const size_t tbytes = (size_t)std::max<uint32_t>(p.info.slab_rows, 1) * b->g.bstride * sizeof(uint4);
    hipError_t e = hipMalloc((void **)&b->d_W, wbytes ? wbytes : 16);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_T, tbytes);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_R, (size_t)b->g.bstride * sizeof(uint4));
    if (e == hipSuccess)
        e = hipMemsetAsync(b->d_R, 0, (size_t)b->g.bstride * sizeof(uint4), b->circ->ctx->stream);
    b->r_known = false;
    return e;
}

gc_batch *gc_batch_create(gc_circ *circ, uint32_t batch, int *status) try {
    int rc = GC_OK;
    gc_batch *b = nullptr;
    if (!circ || batch == 0) rc = GC_E_ARG;
    if (rc == GC_OK) {
        b = new (std::nothrow) gc_batch;
        if (!b) rc = GC_E_NOMEM;
    }
    if (rc == GC_OK) {
        b->circ = circ;
        b->g.batch = batch;
        b->use_graph = std::getenv("GC_NO_GRAPH") == nullptr;  // developer aid: profilers that choke on graph launches
        hipError_t e = hipSetDevice(circ->ctx->device);
        if (e == hipSuccess) e = alloc_buffers(b);
        if (e == hipSuccess) e = hipMalloc((void **)&b->d_rk, 60 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipEventCreate(&b->ev0);
        if (e == hipSuccess) e = hipEventCreate(&b->ev1);
        if (e != hipSuccess) {
            set_error("gc_batch_create", e);
            rc = e == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
        }
    }
    if (rc != GC_OK && b) {
        gc_batch_free(b);
        b = nullptr;
    }
    if (status) *status = rc;
    return b;
} catch (...) {
    const int rc__ = gc::on_exception();
    if (status) *status = rc__;
    return nullptr;
}

static void drop_graphs(gc_batch *b) {
    for (auto &ge : b->graphs) {
        if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
    }
    b->graphs.clear();
}

void gc_batch_free(gc_batch *b) {
    if (!b) return;
    if (b->circ && b->circ->ctx) {
        (void)hipSetDevice(b->circ->ctx->device);
        (void)hipStreamSynchronize(b->circ->ctx->stream);
    }
    drop_graphs(b);
    free_buffers(b);
    if (b->d_rk) (void)hipFree(b->d_rk);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    delete b;
}

uint32_t gc_batch_stride(const gc_batch *b) { return b ? b->g.bstride : 0; }

uint32_t gc_batch_tile_instances(const gc_batch *b) { return !b ? 0 : b->schedule == 0 ? 1u : 1u << b->g.ti_log2; }
int gc_batch_wires_in_lds(const gc_batch *b) { return b && b->schedule != 0 && b->g.lds_wires; }

// the schedule / kernel choice fixes the HBM layout (tile size): when it changes, start over with fresh arrays
// (contents are per-pass anyway)
static int relayout(gc_batch *b) {
    const BatchGeom g = geom_for(b);
    if (b->d_W && g.ti_log2 == b->g.ti_log2 && g.bstride == b->g.bstride && g.ntiles == b->g.ntiles) {
        b->g = g;  // same arrays; lds_wires may differ
        return GC_OK;
    }
    if (std::getenv("GC_TRACE"))
        std::fprintf(stderr, "[gc trace] relayout: batch %u re-sized (ti %u -> %u, stride %u -> %u, tiles %u -> %u, had arrays %d)\n", b->g.batch,
                     b->g.ti_log2, g.ti_log2, b->g.bstride, g.bstride, b->g.ntiles, g.ntiles, b->d_W != nullptr);
    GC_HIP(hipSetDevice(b->circ->ctx->device));
    GC_HIP(hipStreamSynchronize(b->circ->ctx->stream));
    drop_graphs(b);
    free_buffers(b);
    b->timed = false;
    b->have_all_wires = false;
    GC_HIP(alloc_buffers(b));
    return GC_OK;
}

int gc_batch_set_schedule(gc_batch *b, int schedule) try {
    if (!b || schedule < 0 || schedule > 2) return GC_E_ARG;
    b->single_phase = schedule == 2;  // 1 and 2 share the tiled layout; the kernel (and maybe the tile) differs
    b->schedule = schedule ? 1 : 0;
    return relayout(b);
} catch (...) {
    return gc::on_exception();
}

int gc_batch_set_graph(gc_batch *b, int on) {
    if (!b) return GC_E_ARG;
    b->use_graph = on != 0;
    return GC_OK;
}

// The engine's own stream captures (a pass recorded once as a hipGraph and replayed) run one at a time in the process: two
// contexts capturing at once on two threads invalidated each other's capture (hipErrorStreamCaptureInvalidated in one, a
// sticky error in the other: three streams with GC_NO_COOP, tests/test_gpu_stream.py).
static std::mutex &capture_mu() {
    static std::mutex m;
    return m;
}

static int set_key(gc_batch *b, const uint8_t *key, size_t keylen) {
    AesKey k;
    if (!key || !aes_expand_key(key, keylen, &k)) return GC_E_KEYSIZE;
    if (b->rounds == k.rounds && std::memcmp(b->rk_host, k.w, sizeof k.w) == 0) return GC_OK;
    std::memcpy(b->rk_host, k.w, sizeof k.w);
    b->rounds = k.rounds;
    // pageable source: the runtime stages it before returning, so rk_host may change afterwards
    GC_HIP(hipMemcpyAsync(b->d_rk, b->rk_host, sizeof k.w, hipMemcpyHostToDevice, b->circ->ctx->stream));
    return GC_OK;
}

// enqueue the level launches of one garble (eval == false) or eval pass
static void enqueue_levels(gc_batch *b, bool eval, const uint4 *T, hipStream_t s) {
    const Plan &p = b->circ->plan.p;
    LevelArgs a{};
    a.W = b->d_W;
    a.R = b->d_R;
    a.T = const_cast<uint4 *>(T);
    a.rk = b->d_rk;
    a.te0 = b->circ->ctx->d_te0;
    a.rounds = b->rounds;
    for (const Step &st : p.levels) {
        a.descs = b->circ->d_descs + st.first;
        a.count = st.count;
        a.nonfree = st.nonfree;
        a.out_slot0 = p.info.ninputs + st.first;
        if (eval) launch_eval_level(a, b->g, s);
        else launch_garble_level(a, b->g, s);
    }
}

// does this batch run the flattened fused kernels?
static bool uses_flat(const gc_batch *b) {
    return b->schedule == 1 && b->g.lds_wires && want_flat(b) && !b->circ->plan.p.fl_units.empty();
}

// ONE instance of a wide circuit with its wires in HBM (a streamed SSA-step circuit): levels averaging >= 2.5 passes of
// 1024 lanes are spread over workgroups (run_levels)
static bool wide_one_instance(const gc_batch *b, bool eval) {
    const Plan &p = b->circ->plan.p;
    if (b->schedule != 1 || b->g.lds_wires || b->g.batch != 1 || b->d_prof || p.levels.size() < 2) return false;
    uint64_t passes = 0;
    for (const Step &st : p.levels) passes += level1_passes(st, eval);
    return passes * 2 >= (uint64_t)p.levels.size() * 5;
}

static int run_levels(gc_batch *b, bool eval, const uint4 *T, const uint4 *rnd = nullptr) {
    hipStream_t s = b->circ->ctx->stream;
    const Plan &p = b->circ->plan.p;
    // fused, LDS-resident wires, flattened XOR (the production path).  The level-walking kernel below keeps
    // every intermediate wire (store_all / Garbled.Wires).
    if (b->schedule == 1 && b->g.lds_wires && want_flat(b)) {
        FusedFlatArgs f{};
        f.prog = b->circ->d_fl_prog;
        f.units = b->circ->d_fl_units;
        f.hgslot = b->circ->d_fl_hgslot;
        f.ogslot = b->circ->d_fl_ogslot;
        f.in_lds = b->circ->d_fl_in_lds;
        f.nunits = (uint32_t)p.fl_units.size();
        f.ninputs = p.info.ninputs;
        f.nls = p.n_flat_slots;
        f.ustride = p.fl_unit_stride;
        f.W = b->d_W;
        f.R = b->d_R;
        f.T = const_cast<uint4 *>(T);
        f.rk = b->d_rk;
        f.te0 = b->circ->ctx->d_te0;
        f.rounds = b->rounds;
        f.prof = b->d_prof;
        f.rnd = rnd;
        f.has_or = p.info.n_or != 0;
        GC_HIP(launch_fused_flat(eval, f, b->g, s));
        b->last_launches = f.nunits ? 1 : 0;
        b->have_all_wires = false;
        return GC_OK;
    }
    if (b->schedule == 1 && b->g.lds_wires) {  // fused, LDS-resident wires, hash-phase order
        FusedLdsArgs f{};
        f.descs = b->circ->d_fdescs;
        f.gslot = b->circ->d_fgslot;
        f.steps = b->circ->d_fsteps;
        f.in_lds = b->circ->d_in_lds;
        f.nsteps = (uint32_t)p.fsteps.size();
        f.chunks = b->circ->d_fchunks;
        f.nchunks = (uint32_t)p.fchunks.size();
        f.ninputs = p.info.ninputs;
        f.nls = p.n_lds_slots;
        f.W = b->d_W;
        f.R = b->d_R;
        f.T = const_cast<uint4 *>(T);
        f.rk = b->d_rk;
        f.te0 = b->circ->ctx->d_te0;
        f.rounds = b->rounds;
        f.store_all = b->store_all;
        f.prof = b->d_prof;
        GC_HIP(launch_fused_lds(eval, f, b->g, s));
        b->last_launches = f.nsteps ? 1 : 0;
        b->have_all_wires = b->store_all;
        return GC_OK;
    }
    b->have_all_wires = true;
    if (b->schedule == 1) {  // fused with global-memory wires (circuits whose live set exceeds LDS)
        FusedArgs a{};
        a.descs = b->circ->d_descs;
        a.steps = b->circ->d_steps;
        a.nsteps = (uint32_t)p.levels.size();
        a.ninputs = p.info.ninputs;
        a.W = b->d_W;
        a.R = b->d_R;
        a.T = const_cast<uint4 *>(T);
        a.rk = b->d_rk;
        a.te0 = b->circ->ctx->d_te0;
        a.rounds = b->rounds;
        a.prof = b->d_prof;
        a.narrow_col = !std::getenv("GC_NO_COL") && fused_col_form_fits(p.levels.data(), (uint32_t)p.levels.size(), b->g.ti_log2);
        // ONE instance of a wide circuit (a streamed SSA-step circuit): a single workgroup on one CU would walk it alone.
        // When the levels average >= 2.5 passes of 1024 lanes, spread every level's passes over workgroups, one launch
        // per level, recorded once in a hipGraph per (pass, key size, table pointer) and replayed.
        if (wide_one_instance(b, eval)) {
            {
                if (coop_ready(b->circ->ctx)) {  // ONE launch: 32 workgroups of one XCD behind a barrier in its L2
                    gc_ctx *cx = b->circ->ctx;
                    // a workgroup that gives up raises the pinned word *h_coop_err itself (no copy per pass); it stays up
                    // until gc_ctx_coop_check (gc_ctx_sync, gc_pass_dev, the streaming calls that hand results out) notes it
                    // (with its stand-by workgroup: a pass that loses a workgroup is done again inside the same launch)
                    launch_coop(eval, a, cx->d_coop, b->xchg, cx->coop_launched, s);
                    GC_HIP(hipGetLastError());
                    if (a.nsteps) cx->coop_launched++;  // (a launch that really went out) = the device's count of passes once this one has run
                    b->last_launches = 1;
                    return GC_OK;
                }
                b->last_launches = a.nsteps;
                const int kGraphKey = 3;  // distinct from the schedule-0 graphs of this batch
                if (!b->use_graph || b->circ->ctx->capturing || live_contexts() > 1) {
                    launch_levels1(eval, a, p.levels.data(), s);
                    GC_HIP(hipGetLastError());
                    return GC_OK;
                }
                for (auto &ge : b->graphs)
                    if (ge.eval == eval && ge.rounds == b->rounds && ge.T == T && ge.schedule == kGraphKey) {
                        GC_HIP(hipGraphLaunch(ge.exec, s));
                        return GC_OK;
                    }
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                hipError_t e;
                {
                    std::lock_guard<std::mutex> cap(capture_mu());  // (one capture at a time in the process, see capture_mu)
                    e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
                    const hipError_t e_begin = e;
                    hipError_t e_launch = hipSuccess, e_end = hipSuccess;
                    if (e == hipSuccess) {
                        launch_levels1(eval, a, p.levels.data(), s);
                        e_launch = hipGetLastError();
                        e = e_end = hipStreamEndCapture(s, &graph);
                        if (e == hipSuccess) e = e_launch;
                    }
                    if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                    if (e != hipSuccess && std::getenv("GC_TRACE"))
                        std::fprintf(stderr, "[gc trace] level-launch capture failed: begin %d launch %d end %d instantiate %d\n", (int)e_begin,
                                     (int)e_launch, (int)e_end, (int)e);
                }
                if (graph) (void)hipGraphDestroy(graph);
                if (e != hipSuccess) {  // capture unavailable: direct launches of the same kernels
                    (void)hipGetLastError();
                    b->use_graph = false;
                    launch_levels1(eval, a, p.levels.data(), s);
                    GC_HIP(hipGetLastError());
                    return GC_OK;
                }
                b->graphs.push_back({eval, b->rounds, kGraphKey, T, exec});
                GC_HIP(hipGraphLaunch(exec, s));
                return GC_OK;
            }
        }
        if (eval) launch_eval_fused(a, b->g, s);
        else launch_garble_fused(a, b->g, s);
        GC_HIP(hipGetLastError());
        b->last_launches = a.nsteps ? 1 : 0;
        return GC_OK;
    }
    b->last_launches = (uint32_t)p.levels.size();
    if (!b->use_graph || p.levels.size() < 2 || live_contexts() > 1) {
        enqueue_levels(b, eval, T, s);
        GC_HIP(hipGetLastError());
        return GC_OK;
    }
    // one captured graph per (pass, rounds, table pointer, schedule)
    for (auto &ge : b->graphs) {
        if (ge.eval == eval && ge.rounds == b->rounds && ge.T == T && ge.schedule == b->schedule) {
            GC_HIP(hipGraphLaunch(ge.exec, s));
            return GC_OK;
        }
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipError_t e;
    {
        std::lock_guard<std::mutex> cap(capture_mu());
        e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        if (e == hipSuccess) {
            enqueue_levels(b, eval, T, s);
            e = hipStreamEndCapture(s, &graph);
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    }
    if (graph) (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        // capture unsupported in this environment: fall back to direct launches (same kernels)
        (void)hipGetLastError();
        b->use_graph = false;
        enqueue_levels(b, eval, T, s);
        GC_HIP(hipGetLastError());
        return GC_OK;
    }
    b->graphs.push_back({eval, b->rounds, b->schedule, T, exec});
    GC_HIP(hipGraphLaunch(exec, s));
    return GC_OK;
}

int gc_batch_garble(gc_batch *b, const uint8_t *key, size_t keylen, const void *d_rnd) {
    if (!b || !d_rnd) return GC_E_ARG;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    int rc = set_key(b, key, keylen);
    if (rc != GC_OK) return rc;
    b->r_known = false;  // (drawn on the device)
    const Plan &p = b->circ->plan.p;
    // the flattened fused kernel draws R and the input labels itself; the other schedules get them from a small
    // initialisation kernel (outside the timed bracket)
    const bool fused_init = uses_flat(b);
    if (!fused_init) launch_init_garble((const uint4 *)d_rnd, p.info.ninputs, b->d_W, b->d_R, b->g, ctx->stream);
    if (!ctx->capturing) GC_HIP(hipEventRecord(b->ev0, ctx->stream));
    rc = run_levels(b, false, b->d_T, fused_init ? (const uint4 *)d_rnd : nullptr);
    if (rc != GC_OK) return rc;
    if (!ctx->capturing) GC_HIP(hipEventRecord(b->ev1, ctx->stream));
    b->timed = !ctx->capturing;
    return GC_OK;
}

// two batches can exchange labels / tables only if their HBM layouts agree (same schedule class and tile size)
static bool same_layout(const gc_batch *a, const gc_batch *b) {
    return a->circ == b->circ && a->g.batch == b->g.batch && a->schedule == b->schedule &&
           a->g.ti_log2 == b->g.ti_log2 && a->g.bstride == b->g.bstride;
}

int gc_batch_select_inputs(gc_batch *ev, const gc_batch *gb, const void *d_bits) {
    if (!ev || !gb || !d_bits || !same_layout(ev, gb)) return GC_E_ARG;
    gc_ctx *ctx = ev->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    launch_select_inputs(ev->d_W, gb->d_W, gb->d_R, (const uint8_t *)d_bits, ev->circ->plan.p.info.ninputs, ev->g,
                         ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

int gc_batch_set_inputs(gc_batch *ev, const void *d_labels) {
    if (!ev || !d_labels) return GC_E_ARG;
    gc_ctx *ctx = ev->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    uint32_t nin = ev->circ->plan.p.info.ninputs;
    launch_scatter((const uint4 *)d_labels, nin, nin, nullptr, 0, ev->d_W, ev->g.lw, 0, ev->g.batch, ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

int gc_batch_eval(gc_batch *ev, const uint8_t *key, size_t keylen, const gc_batch *tables) {
    if (!ev || !tables || !same_layout(ev, tables)) return GC_E_ARG;
    gc_ctx *ctx = ev->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    int rc = set_key(ev, key, keylen);
    if (rc != GC_OK) return rc;
    if (!ctx->capturing) GC_HIP(hipEventRecord(ev->ev0, ctx->stream));
    rc = run_levels(ev, true, tables->d_T);
    if (rc != GC_OK) return rc;
    if (!ctx->capturing) GC_HIP(hipEventRecord(ev->ev1, ctx->stream));
    ev->timed = !ctx->capturing;
    return GC_OK;
}

int gc_batch_decode(const gc_batch *gb, const gc_batch *ev, void *d_bits_out, void *d_mismatch) {
    if (!gb || !ev || !d_bits_out || !same_layout(gb, ev)) return GC_E_ARG;
    gc_ctx *ctx = gb->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    launch_decode(gb->d_W, gb->d_R, ev->d_W, gb->circ->d_out_slots, gb->circ->plan.p.info.noutputs,
                  (uint8_t *)d_bits_out, (uint32_t *)d_mismatch, gb->g, ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

// ---- read-backs --------------------------------------------------------------------------------

namespace {

// temp device buffer freed on scope exit
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

}  // namespace

int gc_batch_read_r(gc_batch *b, gc_label *r_out) {
    if (!b || !r_out) return GC_E_ARG;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    GC_HIP(hipMemcpyAsync(r_out, b->d_R, (size_t)b->g.batch * sizeof(gc_label), hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GC_OK;
}

// lazily: the copy stream, its events and two device staging buffers of at least `bytes` each
static int ensure_pipeline(gc_ctx *ctx, size_t bytes) {
    if (!ctx->copy_stream) GC_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++) {
        if (!ctx->ev_k[b]) GC_HIP(hipEventCreateWithFlags(&ctx->ev_k[b], hipEventDisableTiming));
        if (!ctx->ev_c[b]) GC_HIP(hipEventCreateWithFlags(&ctx->ev_c[b], hipEventDisableTiming));
    }
    if (bytes > ctx->stage_cap) {
        GC_HIP(hipStreamSynchronize(ctx->stream));
        GC_HIP(hipStreamSynchronize(ctx->copy_stream));
        for (int b = 0; b < 2; b++) {
            if (ctx->stage[b]) (void)hipFree(ctx->stage[b]);
            ctx->stage[b] = nullptr;
        }
        ctx->stage_cap = 0;
        for (int b = 0; b < 2; b++) GC_HIP(hipMalloc(&ctx->stage[b], bytes));
        ctx->stage_cap = bytes;
    }
    return GC_OK;
}

constexpr size_t kPipeChunkBytes = (size_t)32 << 20;  // per staging buffer: long enough for full PCIe rate, short
                                                      // enough that the first chunk's transpose hides nothing big

// device [row][instance] array -> PINNED host [instance][n]: transpose kernel of chunk k + 1 on the ctx stream while the
// DMA of chunk k runs on the copy stream
static int read_gather_pinned(gc_batch *b, const uint4 *src, const Layout &lay, const uint32_t *slots, uint32_t slot0,
                              uint32_t n, int mode, void *host_out) {
    gc_ctx *ctx = b->circ->ctx;
    const size_t elems = (size_t)n * (mode ? 2 : 1), per_inst = elems * sizeof(uint4);
    uint32_t chunk = (uint32_t)std::max<size_t>(32, std::min<size_t>(b->g.batch, kPipeChunkBytes / per_inst));
    chunk = (chunk + 31u) & ~31u;
    int rc = ensure_pipeline(ctx, (size_t)std::min<uint32_t>(chunk, (b->g.batch + 31u) & ~31u) * per_inst);
    if (rc != GC_OK) return rc;
    uint32_t k = 0;
    for (uint32_t i0 = 0; i0 < b->g.batch; i0 += chunk, k++) {
        const int sb = (int)(k & 1);
        const uint32_t cnt = std::min(chunk, b->g.batch - i0);
        if (k >= 2) GC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_c[sb], 0));  // the buffer's previous DMA is done
        launch_gather(src, lay, i0, slots, slot0, n, b->d_R, mode, (uint4 *)ctx->stage[sb], elems, cnt, ctx->stream);
        GC_HIP(hipGetLastError());
        GC_HIP(hipEventRecord(ctx->ev_k[sb], ctx->stream));
        GC_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_k[sb], 0));
        GC_HIP(hipMemcpyAsync((uint8_t *)host_out + (size_t)i0 * per_inst, ctx->stage[sb], (size_t)cnt * per_inst,
                              hipMemcpyDeviceToHost, ctx->copy_stream));
        GC_HIP(hipEventRecord(ctx->ev_c[sb], ctx->copy_stream));
    }
    GC_HIP(hipStreamSynchronize(ctx->copy_stream));
    return GC_OK;
}

static int read_gather(gc_batch *b, const uint4 *src, const Layout &lay, const uint32_t *slots, uint32_t slot0,
                       uint32_t n, int mode, void *host_out) {
    if (n == 0) return GC_OK;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    const size_t elems = (size_t)n * (mode ? 2 : 1);
    if (elems * sizeof(uint4) * b->g.batch >= ((size_t)1 << 20) && gc_host_is_pinned(host_out))
        return read_gather_pinned(b, src, lay, slots, slot0, n, mode, host_out);
    // bounded staging in the ctx's persistent buffer (no hipMalloc / hipFree per call): at most 64 MiB per pass
    const size_t per_inst = elems * sizeof(uint4);
    uint32_t chunk = (uint32_t)std::max<size_t>(32, std::min<size_t>(b->g.batch, ((size_t)64 << 20) / per_inst));
    chunk = (chunk + 31u) & ~31u;
    int rcp = ensure_pipeline(ctx, (size_t)std::min<uint32_t>(chunk, (b->g.batch + 31u) & ~31u) * per_inst);
    if (rcp != GC_OK) return rcp;
    void *tmp = ctx->stage[0];
    for (uint32_t i0 = 0; i0 < b->g.batch; i0 += chunk) {
        const uint32_t cnt = std::min(chunk, b->g.batch - i0);
        launch_gather(src, lay, i0, slots, slot0, n, b->d_R, mode, (uint4 *)tmp, elems, cnt, ctx->stream);
        GC_HIP(hipGetLastError());
        GC_HIP(hipMemcpyAsync((uint8_t *)host_out + (size_t)i0 * per_inst, tmp, (size_t)cnt * per_inst,
                              hipMemcpyDeviceToHost, ctx->stream));
        GC_HIP(hipStreamSynchronize(ctx->stream));
    }
    return GC_OK;
}

int gc_batch_read_slab(gc_batch *b, gc_label *slab_out) {
    if (!b || !slab_out) return GC_E_ARG;
    return read_gather(b, b->d_T, b->g.lt, nullptr, 0, b->circ->plan.p.info.slab_rows, 0, slab_out);
}

int gc_batch_set_store_all(gc_batch *b, int on) try {
    if (!b) return GC_E_ARG;
    b->store_all = on != 0;
    return relayout(b);
} catch (...) {
    return gc::on_exception();
}

int gc_batch_read_wires(gc_batch *b, gc_wire *wires_out) {
    if (!b || !wires_out || !b->have_all_wires) return GC_E_ARG;
    return read_gather(b, b->d_W, b->g.lw, b->circ->d_slot_of_wire, 0, b->circ->plan.p.info.nwires, 1, wires_out);
}

int gc_batch_read_labels(gc_batch *b, gc_label *labels_out) {
    if (!b || !labels_out || !b->have_all_wires) return GC_E_ARG;
    return read_gather(b, b->d_W, b->g.lw, b->circ->d_slot_of_wire, 0, b->circ->plan.p.info.nwires, 0, labels_out);
}

int gc_batch_read_outputs(gc_batch *b, gc_label *out) {
    if (!b || !out) return GC_E_ARG;
    return read_gather(b, b->d_W, b->g.lw, b->circ->d_out_slots, 0, b->circ->plan.p.info.noutputs, 0, out);
}

// PINNED host [instance][src_stride] -> device [row][instance]: DMA of chunk k + 1 on the copy stream while the
// scatter kernel of chunk k runs on the ctx stream
static int write_scatter_pinned(gc_batch *b, const void *host_src, size_t src_stride_elems, size_t col0, uint32_t n,
                                const uint32_t *slots, uint32_t slot0, uint4 *dst, const Layout &lay) {
    gc_ctx *ctx = b->circ->ctx;
    const size_t per_inst = (size_t)n * sizeof(uint4);
    uint32_t chunk = (uint32_t)std::max<size_t>(32, std::min<size_t>(b->g.batch, kPipeChunkBytes / per_inst));
    chunk = (chunk + 31u) & ~31u;
    int rc = ensure_pipeline(ctx, (size_t)std::min<uint32_t>(chunk, (b->g.batch + 31u) & ~31u) * per_inst);
    if (rc != GC_OK) return rc;
    // what is already queued on the ctx stream may still read the staging buffers (an earlier pipelined call)
    GC_HIP(hipEventRecord(ctx->ev_k[0], ctx->stream));
    GC_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_k[0], 0));
    uint32_t k = 0;
    for (uint32_t i0 = 0; i0 < b->g.batch; i0 += chunk, k++) {
        const int sb = (int)(k & 1);
        const uint32_t cnt = std::min(chunk, b->g.batch - i0);
        const uint8_t *src = (const uint8_t *)host_src + ((size_t)i0 * src_stride_elems + col0) * sizeof(uint4);
        if (k >= 2) GC_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_k[sb], 0));  // its previous scatter is done
        if (src_stride_elems == n)
            GC_HIP(hipMemcpyAsync(ctx->stage[sb], src, (size_t)cnt * per_inst, hipMemcpyHostToDevice, ctx->copy_stream));
        else
            GC_HIP(hipMemcpy2DAsync(ctx->stage[sb], per_inst, src, src_stride_elems * sizeof(uint4), per_inst, cnt,
                                    hipMemcpyHostToDevice, ctx->copy_stream));
        GC_HIP(hipEventRecord(ctx->ev_c[sb], ctx->copy_stream));
        GC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_c[sb], 0));
        launch_scatter((const uint4 *)ctx->stage[sb], n, n, slots, slot0, dst, lay, i0, cnt, ctx->stream);
        GC_HIP(hipGetLastError());
        GC_HIP(hipEventRecord(ctx->ev_k[sb], ctx->stream));
    }
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GC_OK;
}

static int write_scatter(gc_batch *b, const void *host_src, size_t src_stride_elems, size_t col0, uint32_t n,
                         const uint32_t *slots, uint32_t slot0, uint4 *dst, const Layout &lay) {
    if (n == 0) return GC_OK;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    const size_t per_inst = (size_t)n * sizeof(uint4);
    if (per_inst * b->g.batch >= ((size_t)1 << 20) && gc_host_is_pinned(host_src))
        return write_scatter_pinned(b, host_src, src_stride_elems, col0, n, slots, slot0, dst, lay);
    uint32_t chunk = (uint32_t)std::max<size_t>(32, std::min<size_t>(b->g.batch, ((size_t)64 << 20) / per_inst));
    chunk = (chunk + 31u) & ~31u;
    int rcp = ensure_pipeline(ctx, (size_t)std::min<uint32_t>(chunk, (b->g.batch + 31u) & ~31u) * per_inst);
    if (rcp != GC_OK) return rcp;
    void *tmp = ctx->stage[0];
    for (uint32_t i0 = 0; i0 < b->g.batch; i0 += chunk) {
        const uint32_t cnt = std::min(chunk, b->g.batch - i0);
        const uint8_t *src = (const uint8_t *)host_src + ((size_t)i0 * src_stride_elems + col0) * sizeof(uint4);
        if (src_stride_elems == n)
            GC_HIP(hipMemcpyAsync(tmp, src, (size_t)cnt * per_inst, hipMemcpyHostToDevice, ctx->stream));
        else
            GC_HIP(hipMemcpy2DAsync(tmp, per_inst, src, src_stride_elems * sizeof(uint4), per_inst, cnt,
                                    hipMemcpyHostToDevice, ctx->stream));
        launch_scatter((const uint4 *)tmp, n, n, slots, slot0, dst, lay, i0, cnt, ctx->stream);
        GC_HIP(hipGetLastError());
        GC_HIP(hipStreamSynchronize(ctx->stream));
    }
    return GC_OK;
}

int gc_batch_write_slab(gc_batch *b, const gc_label *slab) {
    if (!b || !slab) return GC_E_ARG;
    uint32_t rows = b->circ->plan.p.info.slab_rows;
    return write_scatter(b, slab, rows, 0, rows, nullptr, 0, b->d_T, b->g.lt);
}

void *gc_batch_dev_wires(gc_batch *b) { return b ? b->d_W : nullptr; }
void *gc_batch_dev_slab(gc_batch *b) { return b ? b->d_T : nullptr; }
void *gc_batch_dev_r(gc_batch *b) { return b ? b->d_R : nullptr; }

int gc_batch_gather_outputs(gc_batch *b, void *d_out) {
    if (!b || !d_out) return GC_E_ARG;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    launch_gather_rows(b->d_W, b->circ->d_out_slots, b->circ->plan.p.info.noutputs, (uint4 *)d_out, b->g,
                       ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

// garbler: both labels of input wires [first, first + count) of every instance, as gc_wire {L0, L0^R} in a dense
// device buffer [batch][count] — what the OT sender (COT) consumes
int gc_batch_gather_input_wires(gc_batch *b, uint32_t first, uint32_t count, void *d_wires_out) {
    if (!b || (!d_wires_out && count)) return GC_E_ARG;
    const uint32_t nin = b->circ->plan.p.info.ninputs;
    if (first > nin || count > nin - first) return GC_E_ARG;
    if (count == 0) return GC_OK;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    launch_gather(b->d_W, b->g.lw, 0, nullptr, first, count, b->d_R, 1, (uint4 *)d_wires_out, 2 * (size_t)count,
                  b->g.batch, ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

// evaluator: active labels of input wires [first, first + count) from a dense device buffer gc_label [batch][count]
int gc_batch_set_input_range(gc_batch *ev, uint32_t first, uint32_t count, const void *d_labels) {
    if (!ev || (!d_labels && count)) return GC_E_ARG;
    const uint32_t nin = ev->circ->plan.p.info.ninputs;
    if (first > nin || count > nin - first) return GC_E_ARG;
    if (count == 0) return GC_OK;
    gc_ctx *ctx = ev->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    launch_scatter((const uint4 *)d_labels, count, count, nullptr, first, ev->d_W, ev->g.lw, 0, ev->g.batch, ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

int gc_batch_debug_profile(gc_batch *b, int enable, uint64_t *out8) try {
    if (!b) return GC_E_ARG;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    const size_t n = (size_t)b->g.ntiles * 16;
    if (out8 && b->d_prof) {  // average over workgroups (16 counters per workgroup)
        std::vector<uint64_t> h(n);
        GC_HIP(hipMemcpy(h.data(), b->d_prof, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (int k = 0; k < 16; k++) {
            unsigned __int128 acc = 0;
            for (uint32_t t = 0; t < b->g.ntiles; t++) acc += h[(size_t)t * 16 + k];
            out8[k] = (uint64_t)(acc / b->g.ntiles);
        }
    }
    if (enable && !b->d_prof) {
        GC_HIP(hipMalloc((void **)&b->d_prof, n * sizeof(uint64_t)));
        GC_HIP(hipMemset(b->d_prof, 0, n * sizeof(uint64_t)));
    } else if (!enable && b->d_prof) {
        (void)hipFree(b->d_prof);
        b->d_prof = nullptr;
    }
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

size_t gc_tables_wire_bytes(const gc_circ *c) {
    if (!c) return 0;
    const gc_plan_info &i = c->plan.p.info;
    return 4 + 4 * (size_t)i.ngates + 16 * (size_t)i.slab_rows;
}

int gc_batch_egress_tables(gc_batch *b, void *d_out, size_t stride) {
    if (!b || !d_out || stride < gc_tables_wire_bytes(b->circ) || (stride & 3)) return GC_E_ARG;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    launch_tables_egress(b->d_T, b->g.lt, b->circ->d_ops, b->circ->d_row_of_gate, b->circ->plan.p.info.ngates,
                         b->g.batch, (uint8_t *)d_out, stride, ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

int gc_batch_ingest_tables(gc_batch *b, const void *d_in, size_t stride, void *d_bad) {
    if (!b || !d_in || !d_bad || stride < gc_tables_wire_bytes(b->circ) || (stride & 3)) return GC_E_ARG;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    launch_tables_ingest(b->d_T, b->g.lt, b->circ->d_ops, b->circ->d_row_of_gate, b->circ->plan.p.info.ngates,
                         b->g.batch, (const uint8_t *)d_in, stride, (uint32_t *)d_bad, ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

int gc_batch_egress_tables_dense(gc_batch *b, void *d_out, size_t stride) {
    if (!b || !d_out || (stride & 15) || stride < (size_t)b->circ->plan.p.info.slab_rows * 16) return GC_E_ARG;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    launch_slab_be(b->d_T, b->g.lt, b->circ->plan.p.info.slab_rows, b->g.batch, (uint8_t *)d_out, stride, false,
                   ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

int gc_batch_ingest_tables_dense(gc_batch *b, const void *d_in, size_t stride) {
    if (!b || !d_in || (stride & 15) || stride < (size_t)b->circ->plan.p.info.slab_rows * 16) return GC_E_ARG;
    gc_ctx *ctx = b->circ->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    launch_slab_be(b->d_T, b->g.lt, b->circ->plan.p.info.slab_rows, b->g.batch, (uint8_t *)const_cast<void *>(d_in), stride,
                   true, ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

float gc_batch_last_ms(gc_batch *b) {
    if (!b || !b->timed) return -1.0f;
    if (hipSetDevice(b->circ->ctx->device) != hipSuccess) return -1.0f;
    if (hipEventSynchronize(b->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, b->ev0, b->ev1) != hipSuccess) return -1.0f;
    return ms;
}

uint32_t gc_batch_last_launches(gc_batch *b) { return b ? b->last_launches : 0; }

// ---- host-buffer API -------------------------------------------------------------------------

// R of a one-instance pass from the host: a stream garbles every circuit under the same R, and the pooled batch it gets
// back usually holds it already (a copy per pass is a blit kernel plus its gaps on the stream)
static hipError_t upload_r(gc_batch *b, const gc_label *r) {
    if (b->r_known && std::memcmp(&b->r_host, r, sizeof(gc_label)) == 0) return hipSuccess;
    b->r_known = false;
    hipError_t e = hipMemcpyAsync(b->d_R, r, sizeof(gc_label), hipMemcpyHostToDevice, b->circ->ctx->stream);
    if (e == hipSuccess) {
        b->r_host = *r;
        b->r_known = true;
    }
    return e;
}

static gc_batch *pool_get(gc_circ *c, uint32_t batch, int *rc) {
    int schedule;
    bool single_phase;
    {
        std::lock_guard<std::mutex> lk(c->pool_mu);
        schedule = c->schedule;
        single_phase = c->single_phase;
        for (size_t i = 0; i < c->pool.size(); i++) {
            if (c->pool[i]->g.batch == batch && c->pool[i]->schedule == schedule) {
                gc_batch *b = c->pool[i];
                c->pool.erase(c->pool.begin() + (long)i);
                if (b->single_phase != single_phase) {
                    b->single_phase = single_phase;
                    if (relayout(b) != GC_OK) {
                        gc_batch_free(b);
                        if (rc) *rc = GC_E_HIP;
                        return nullptr;
                    }
                }
                return b;
            }
        }
    }
    if (std::getenv("GC_TRACE"))
        std::fprintf(stderr, "[gc trace] pool_get: NEW batch of %u for a circuit of %u gates (pool holds %zu, schedule %d)\n", batch,
                     c->plan.p.info.ngates, c->pool.size(), schedule);
    gc_batch *b = gc_batch_create(c, batch, rc);
    if (b && (b->schedule != schedule || b->single_phase != single_phase)) {
        int r2 = gc_batch_set_schedule(b, schedule == 0 ? 0 : single_phase ? 2 : 1);
        if (r2 != GC_OK) {
            gc_batch_free(b);
            if (rc) *rc = r2;
            return nullptr;
        }
    }
    return b;
}

static void pool_put(gc_circ *c, gc_batch *b) {
    std::lock_guard<std::mutex> lk(c->pool_mu);
    if (c->pool.size() < 4) {
        c->pool.push_back(b);
        return;
    }
    gc_batch_free(b);
}

int gc_garble(gc_circ *c, const uint8_t *key, size_t keylen, const uint8_t *rnd, size_t rndlen, uint32_t batch,
              gc_label *r_out, gc_wire *wires_out, gc_wire *io_out, gc_label *slab_out) try {
    if (!c || !rnd || batch == 0) return GC_E_ARG;
    const Plan &p = c->plan.p;
    // error order of the reference: R is read first (garble.go:253), then aes.NewCipher (:260),
    // then the input labels (:271-278)
    if (rndlen < 16) return GC_E_RAND;
    AesKey k;
    if (!key || !aes_expand_key(key, keylen, &k)) return GC_E_KEYSIZE;
    const size_t stride = 16 * ((size_t)p.info.ninputs + 1);
    if (rndlen < stride * batch) return GC_E_RAND;
    int rc = GC_OK;
    Trace tr("gc_garble");
    gc_batch *b = pool_get(c, batch, &rc);
    if (!b) return rc;
    gc_ctx *ctx = c->ctx;
    // one HIP stream per ctx: serialise host-buffer calls that share the ctx
    std::lock_guard<std::mutex> lk(ctx->mu);
    tr.lap("pool + lock");
    do {
        DevBuf d_rnd;
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = d_rnd.alloc(stride * batch);
        if (e == hipSuccess) e = hipMemcpyAsync(d_rnd.p, rnd, stride * batch, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            set_error("gc_garble", e);
            rc = e == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
            break;
        }
        // a pooled batch keeps the geometry of its last use: keeping every wire runs the level-walking kernel, whose
        // tile may be smaller than the flattened kernels' (relayout re-sizes the arrays if it is)
        b->store_all = wires_out != nullptr;
        if ((rc = relayout(b)) != GC_OK) break;
        tr.lap("rnd h2d + layout");
        rc = gc_batch_garble(b, key, keylen, d_rnd.p);
        if (rc != GC_OK) break;
        if (r_out && (rc = gc_batch_read_r(b, r_out)) != GC_OK) break;
        tr.lap("garble + R");
        if (slab_out && (rc = gc_batch_read_slab(b, slab_out)) != GC_OK) break;
        tr.lap("slab d2h");
        if (wires_out && (rc = gc_batch_read_wires(b, wires_out)) != GC_OK) break;
        if (io_out) {
            // Wires[0:ninputs] then Wires[nwires-noutputs:]: both ranges gathered on the device into one dense
            // [batch][ninputs + noutputs] wire array (two launches, column offset), one copy to the caller
            const uint32_t nio = p.info.ninputs + p.info.noutputs;
            const size_t bytes = (size_t)batch * nio * sizeof(gc_wire);
            if ((rc = ensure_pipeline(ctx, std::max(bytes, (size_t)1 << 20))) != GC_OK) break;
            uint4 *st = (uint4 *)ctx->stage[0];
            // a wire = 2 labels: row stride 2 * nio elements, the output range starts 2 * ninputs elements in
            launch_gather(b->d_W, b->g.lw, 0, nullptr, 0, p.info.ninputs, b->d_R, 1, st, 2 * (size_t)nio, batch, ctx->stream);
            launch_gather(b->d_W, b->g.lw, 0, c->d_out_slots, 0, p.info.noutputs, b->d_R, 1, st + 2 * (size_t)p.info.ninputs,
                          2 * (size_t)nio, batch, ctx->stream);
            hipError_t eg = hipGetLastError();
            if (eg == hipSuccess) eg = hipMemcpyAsync(io_out, st, bytes, hipMemcpyDeviceToHost, ctx->stream);
            if (eg == hipSuccess) eg = hipStreamSynchronize(ctx->stream);
            if (eg != hipSuccess) {
                set_error("gc_garble (io wires)", eg);
                rc = GC_E_HIP;
                break;
            }
        }
        tr.lap("wires / io d2h");
        hipError_t es = hipStreamSynchronize(ctx->stream);
        if (es != hipSuccess) {
            set_error("gc_garble", es);
            rc = GC_E_HIP;
        }
    } while (0);
    tr.lap("sync + frees");
    pool_put(c, b);
    return rc;
} catch (...) {
    return gc::on_exception();
}

}  // extern "C"

// One instance garbled from explicit labels; the pooled batch is handed to the caller with the tables still in
// b->d_T (the streaming garbler serialises them on the device) and goes back with gc_circ_release_batch.
int gc_garble_labels_keep(gc_circ *c, const uint8_t *key, size_t keylen, const gc_label *r, const gc_label *inputs,
                          gc_label *out_l0, gc_batch **bout) {
    if (!c || !r || !bout) return GC_E_ARG;
    *bout = nullptr;
    const Plan &p = c->plan.p;
    if (p.info.ninputs && !inputs) return GC_E_ARG;
    int rc = GC_OK;
    gc_batch *b = pool_get(c, 1, &rc);
    if (!b) return rc;
    gc_ctx *ctx = c->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    do {
        hipError_t e = hipSetDevice(ctx->device);
        b->store_all = false;
        if (e == hipSuccess && (rc = relayout(b)) != GC_OK) break;  // before anything is written into the arrays
        if (e == hipSuccess) e = upload_r(b, r);
        if (e != hipSuccess) {
            set_error("gc_garble_labels", e);
            rc = GC_E_HIP;
            break;
        }
        if (p.info.ninputs) rc = write_scatter(b, inputs, p.info.ninputs, 0, p.info.ninputs, nullptr, 0, b->d_W, b->g.lw);
        if (rc != GC_OK) break;
        if ((rc = set_key(b, key, keylen)) != GC_OK) break;
        if ((rc = run_levels(b, false, b->d_T)) != GC_OK) break;
        if (out_l0 && (rc = gc_batch_read_outputs(b, out_l0)) != GC_OK) break;
    } while (0);
    if (rc != GC_OK) {
        pool_put(c, b);
        return rc;
    }
    *bout = b;
    return GC_OK;
}

bool gc_circ_flat_poll(gc_circ *c, bool start) {
    if (c->flat_ready.load(std::memory_order_acquire)) return true;
    int idle = 0;
    if (start && c->flat_async.compare_exchange_strong(idle, 1)) {
        try {
            c->flat_thread = std::thread([c] {
                (void)hipSetDevice(c->ctx->device);
                circ_ensure_flat(c);
            });
        } catch (...) {  // no thread: the next caller that needs the plan builds it itself
            c->flat_async.store(0);
        }
    }
    return false;
}
bool gc_circ_flat_job(gc_circ *c, gc::FlatJob *j, size_t *lds_bytes, bool *has_or) {
    if (!c || !j) return false;
    circ_ensure_flat(c);
    const Plan &p = c->plan.p;
    if (p.n_flat_slots == 0xffffffffu || p.fl_units.empty() || p.info.slab_rows >= (1u << 22) || !c->d_fl_prog) return false;
    const size_t need = fused_flat_bytes(p.n_flat_slots, 0, p.fl_unit_stride);
    if (need > (size_t)160 * 1024) return false;
    *j = FlatJob{};
    j->prog = (const uint4 *)c->d_fl_prog;
    j->units = c->d_fl_units;
    j->hgslot = c->d_fl_hgslot;
    j->ogslot = c->d_fl_ogslot;
    j->in_lds = c->d_fl_in_lds;
    j->nunits = (uint32_t)p.fl_units.size();
    j->ninputs = p.info.ninputs;
    j->ti_log2 = 0;
    j->zslot = p.n_flat_slots - 1;
    j->ustride = p.fl_unit_stride;
    j->w_tile = p.info.nslots;
    j->t_tile = std::max<uint32_t>(p.info.slab_rows, 1);
    j->te0 = c->ctx->d_te0;
    j->batch = 1;
    if (lds_bytes) *lds_bytes = need;
    if (has_or) *has_or = p.info.n_or != 0;
    return true;
}

void gc_circ_release_batch(gc_circ *c, gc_batch *b) {
    if (c && b) pool_put(c, b);
}

int gc_pass_dev(gc_circ *c, bool eval, const uint8_t *key, size_t keylen, const gc_label *r, const void *d_store,
                const uint32_t *d_in_idx, const uint32_t *d_out_idx, const gc_label *slab_host, size_t slab_rows,
                gc_batch **bout, const gc::StoreXchg *d_xchg) {
    if (!c || !bout || !d_store || (!eval && !r)) return GC_E_ARG;
    // *bout set (gc_pass_batch): the caller took the batch beforehand and, as evaluator, has put (or is putting, on
    // another stream the ctx stream waits for) the tables into b->d_T itself — slab_host is null then
    gc_batch *b = *bout;
    *bout = nullptr;
    const Plan &p = c->plan.p;
    if (b && (b->circ != c || b->g.batch != 1)) return GC_E_ARG;
    if (eval && (slab_rows != p.info.slab_rows || (!slab_host && slab_rows && !b))) {
        if (b) pool_put(c, b);
        return GC_E_ROWS;
    }
    int rc = GC_OK;
    if (!b) b = pool_get(c, 1, &rc);
    if (!b) return rc;
    gc_ctx *ctx = c->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    do {
        hipError_t e = hipSetDevice(ctx->device);
        b->store_all = false;
        if ((rc = gc_ctx_coop_check(ctx)) != GC_OK) break;
        if (e == hipSuccess && (rc = relayout(b)) != GC_OK) break;
        if (e == hipSuccess && !eval) e = upload_r(b, r);
        // one instance: a tile of one, the table array is the dense slab [row] and wire slot w is element w
        if (e == hipSuccess && eval && slab_rows && slab_host)
            e = hipMemcpyAsync(b->d_T, slab_host, slab_rows * sizeof(gc_label), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            set_error("gc_pass_dev", e);
            rc = GC_E_HIP;
            break;
        }
        // a cooperative pass exchanges the labels with the store itself (a kernel before and one behind it were 10 us
        // plus their gaps of a 230 us step)
        // (the caller has put the same pointers into *d_xchg, in device memory, with its other uploads)
        const bool in_pass = d_xchg && wide_one_instance(b, eval) && coop_ready(ctx);
        if (!in_pass) launch_store_gather(b->d_W, (const uint4 *)d_store, d_in_idx, p.info.ninputs, ctx->stream);
        if ((rc = set_key(b, key, keylen)) != GC_OK) break;
        b->xchg = in_pass ? d_xchg : nullptr;
        rc = run_levels(b, eval, b->d_T);
        b->xchg = nullptr;
        if (rc != GC_OK) break;
        if (!in_pass)
            launch_store_scatter((uint4 *)const_cast<void *>(d_store), b->d_W, c->d_out_slots, d_out_idx, p.info.noutputs, ctx->stream);
        e = hipGetLastError();
        if (e != hipSuccess) {
            set_error("gc_pass_dev", e);
            rc = GC_E_HIP;
        }
    } while (0);
    if (rc != GC_OK) {
        pool_put(c, b);
        return rc;
    }
    *bout = b;
    return GC_OK;
}

int gc_pass_batch(gc_circ *c, gc_batch **bout) {
    if (!c || !bout) return GC_E_ARG;
    int rc = GC_OK;
    *bout = pool_get(c, 1, &rc);
    if (!*bout) return rc;
    std::lock_guard<std::mutex> lk(c->ctx->mu);
    if (hipSetDevice(c->ctx->device) != hipSuccess || (rc = relayout(*bout)) != GC_OK) {
        pool_put(c, *bout);
        *bout = nullptr;
        return rc != GC_OK ? rc : GC_E_HIP;
    }
    return GC_OK;
}

extern "C" {

int gc_garble_labels(gc_circ *c, const uint8_t *key, size_t keylen, const gc_label *r, const gc_label *inputs,
                     gc_label *slab_out, gc_label *out_l0) try {
    gc_batch *b = nullptr;
    int rc = gc_garble_labels_keep(c, key, keylen, r, inputs, out_l0, &b);
    if (rc != GC_OK) return rc;
    if (slab_out) {
        std::lock_guard<std::mutex> lk(c->ctx->mu);
        rc = gc_batch_read_slab(b, slab_out);
    }
    pool_put(c, b);
    return rc;
} catch (...) {
    return gc::on_exception();
}

int gc_eval(gc_circ *c, const uint8_t *key, size_t keylen, uint32_t batch, gc_label *wires_inout,
            const gc_label *inputs, const gc_label *slab, size_t slab_rows_given, gc_label *out_labels) try {
    if (!c || batch == 0 || (!wires_inout && !inputs)) return GC_E_ARG;
    const Plan &p = c->plan.p;
    AesKey k;
    if (!key || !aes_expand_key(key, keylen, &k)) return GC_E_KEYSIZE;
    if (slab_rows_given != p.info.slab_rows || (!slab && p.info.slab_rows)) return GC_E_ROWS;
    int rc = GC_OK;
    Trace tr("gc_eval");
    gc_batch *b = pool_get(c, batch, &rc);
    if (!b) return rc;
    gc_ctx *ctx = c->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    do {
        b->store_all = wires_inout != nullptr;
        if ((rc = relayout(b)) != GC_OK) break;  // before the tables and labels go into the arrays
        if (p.info.slab_rows && (rc = gc_batch_write_slab(b, slab)) != GC_OK) break;
        tr.lap("slab h2d");
        if (wires_inout)
            rc = write_scatter(b, wires_inout, p.info.nwires, 0, p.info.ninputs, nullptr, 0, b->d_W, b->g.lw);
        else
            rc = write_scatter(b, inputs, p.info.ninputs, 0, p.info.ninputs, nullptr, 0, b->d_W, b->g.lw);
        if (rc != GC_OK) break;
        tr.lap("inputs h2d");
        rc = gc_batch_eval(b, key, keylen, b);
        if (rc != GC_OK) break;
        if (wires_inout && (rc = gc_batch_read_labels(b, wires_inout)) != GC_OK) break;
        if (out_labels && (rc = gc_batch_read_outputs(b, out_labels)) != GC_OK) break;
        hipError_t es = hipStreamSynchronize(ctx->stream);
        tr.lap("eval + outputs");
        if (es != hipSuccess) {
            set_error("gc_eval", es);
            rc = GC_E_HIP;
        }
    } while (0);
    pool_put(c, b);
    return rc;
} catch (...) {
    return gc::on_exception();
}


// ---- host-buffer calls that speak the 2-party driver's wire format (SURVEY §8f row 1 on the host API) ------------
// gc_garble_wire = Circuit.Garble + the table send loop of circuit.Garbler (garbler.go:53-82): the tables leave the
// device already serialised (k_tables_egress), so the host does ONE conn.Write per instance instead of a
// SendUint32 + SendLabel loop over 36 663 gates.  gc_eval_wire = the receive loop of circuit.Evaluator
// (evaluator.go:40-66) + Circuit.Eval.

int gc_garble_wire(gc_circ *c, const uint8_t *key, size_t keylen, const uint8_t *rnd, size_t rndlen, uint32_t batch,
                   gc_label *r_out, gc_wire *io_out, uint8_t *wire_out, size_t stride) try {
    if (!c || !rnd || batch == 0 || !wire_out) return GC_E_ARG;
    const Plan &p = c->plan.p;
    const size_t need = gc_tables_wire_bytes(c);
    if (stride < need || (stride & 3)) return GC_E_ARG;
    if (rndlen < 16) return GC_E_RAND;
    AesKey k;
    if (!key || !aes_expand_key(key, keylen, &k)) return GC_E_KEYSIZE;
    const size_t rstride = 16 * ((size_t)p.info.ninputs + 1);
    if (rndlen < rstride * batch) return GC_E_RAND;
    int rc = GC_OK;
    gc_batch *b = pool_get(c, batch, &rc);
    if (!b) return rc;
    gc_ctx *ctx = c->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    do {
        DevBuf d_rnd, d_wire;
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = d_rnd.alloc(rstride * batch);
        if (e == hipSuccess) e = d_wire.alloc(stride * batch);
        if (e == hipSuccess) e = hipMemcpyAsync(d_rnd.p, rnd, rstride * batch, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            set_error("gc_garble_wire", e);
            rc = e == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
            break;
        }
        b->store_all = false;
        if ((rc = relayout(b)) != GC_OK) break;
        if ((rc = gc_batch_garble(b, key, keylen, d_rnd.p)) != GC_OK) break;
        if ((rc = gc_batch_egress_tables(b, d_wire.p, stride)) != GC_OK) break;
        e = hipMemcpyAsync(wire_out, d_wire.p, stride * batch, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) {
            set_error("gc_garble_wire", e);
            rc = GC_E_HIP;
            break;
        }
        if (r_out && (rc = gc_batch_read_r(b, r_out)) != GC_OK) break;
        if (io_out) {
            const uint32_t nio = p.info.ninputs + p.info.noutputs;
            const size_t bytes = (size_t)batch * nio * sizeof(gc_wire);
            if ((rc = ensure_pipeline(ctx, std::max(bytes, (size_t)1 << 20))) != GC_OK) break;
            uint4 *st = (uint4 *)ctx->stage[0];
            launch_gather(b->d_W, b->g.lw, 0, nullptr, 0, p.info.ninputs, b->d_R, 1, st, 2 * (size_t)nio, batch, ctx->stream);
            launch_gather(b->d_W, b->g.lw, 0, c->d_out_slots, 0, p.info.noutputs, b->d_R, 1, st + 2 * (size_t)p.info.ninputs,
                          2 * (size_t)nio, batch, ctx->stream);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(io_out, st, bytes, hipMemcpyDeviceToHost, ctx->stream);
            if (e != hipSuccess) {
                set_error("gc_garble_wire (io wires)", e);
                rc = GC_E_HIP;
                break;
            }
        }
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("gc_garble_wire", e);
            rc = GC_E_HIP;
        }
    } while (0);
    pool_put(c, b);
    return rc;
} catch (...) {
    return gc::on_exception();
}

int gc_eval_wire(gc_circ *c, const uint8_t *key, size_t keylen, uint32_t batch, const gc_label *inputs,
                 const uint8_t *wire_in, size_t stride, gc_label *out_labels, uint32_t *bad) try {
    if (!c || batch == 0 || !inputs || !wire_in || !bad) return GC_E_ARG;
    const Plan &p = c->plan.p;
    if (stride < gc_tables_wire_bytes(c) || (stride & 3)) return GC_E_ARG;
    AesKey k;
    if (!key || !aes_expand_key(key, keylen, &k)) return GC_E_KEYSIZE;
    int rc = GC_OK;
    gc_batch *b = pool_get(c, batch, &rc);
    if (!b) return rc;
    gc_ctx *ctx = c->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    do {
        DevBuf d_wire, d_bad;
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = d_wire.alloc(stride * batch);
        if (e == hipSuccess) e = d_bad.alloc(sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(d_bad.p, 0, sizeof(uint32_t), ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_wire.p, wire_in, stride * batch, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            set_error("gc_eval_wire", e);
            rc = e == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
            break;
        }
        b->store_all = false;
        if ((rc = relayout(b)) != GC_OK) break;
        if ((rc = gc_batch_ingest_tables(b, d_wire.p, stride, d_bad.p)) != GC_OK) break;
        e = hipMemcpyAsync(bad, d_bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("gc_eval_wire", e);
            rc = GC_E_HIP;
            break;
        }
        if (*bad) {  // "wrong number of gates" (evaluator.go:44-47) / a row count Eval would reject (eval.go:54-56,86-89)
            rc = GC_E_ROWS;
            break;
        }
        if ((rc = write_scatter(b, inputs, p.info.ninputs, 0, p.info.ninputs, nullptr, 0, b->d_W, b->g.lw)) != GC_OK) break;
        if ((rc = gc_batch_eval(b, key, keylen, b)) != GC_OK) break;
        if (out_labels && (rc = gc_batch_read_outputs(b, out_labels)) != GC_OK) break;
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("gc_eval_wire", e);
            rc = GC_E_HIP;
        }
    } while (0);
    pool_put(c, b);
    return rc;
} catch (...) {
    return gc::on_exception();
}

}  // extern "C"
