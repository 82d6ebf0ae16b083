// comm.cpp — multi-GPU entry points of the C ABI (gc_comm_*): the ONE exchange step of the path, the gather of the
// decoded outputs / output labels of independent instance shards, on RCCL over xGMI.
//
// The reference has no multi-device code; instances are independent (fresh R and labels per Garble call,
// circuit/garble.go:253-278), so every device garbles / evaluates a contiguous instance range with its own gc_ctx and the
// only collective is this terminal ncclAllGather.  Two ways to form a communicator, both reachable from a Go host:
//   * one process (or OS thread) per GPU:  gc_comm_get_unique_id on rank 0, the host distributes the 128 bytes over
//     whatever control channel it has (apps/garbled: its p2p.Conn; bench.py: the launcher's store), every rank calls
//     gc_comm_init_rank;
//   * one process, one gc_ctx per device:   gc_comm_init_all (ncclCommInitAll) and the *_all calls, which bracket the
//     per-device enqueues with ncclGroupStart / ncclGroupEnd.
// librccl is opened lazily (dlopen "librccl.so.1"; a copy the process already holds — e.g. PyTorch's — is reused, its
// soname is the same), so libgcengine.so itself loads on hosts without RCCL and the single-GPU path never maps it.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "engine.h"
#include "rccl_binding.h"

namespace {

using gc_rccl::Rccl;

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *cands[] = {std::getenv("GC_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *c : cands) {
            if (!c || !*c) continue;
            r.h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
            std::snprintf(r.why, sizeof r.why, "%s", dlerror());
        }
        if (!r.h) return;
        bool all = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(r.h, name);
            if (!p) {
                all = false;
                std::snprintf(r.why, sizeof r.why, "librccl lacks %s", name);
            }
            return p;
        };
        r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.ok = all;
    });
    return r;
}

int rccl_missing() {
    std::snprintf(gc::tls_error, sizeof gc::tls_error, "RCCL unavailable: %s", rccl().why[0] ? rccl().why : "librccl.so.1 not found");
    return GC_E_HIP;
}

int rccl_fail(const char *what, ncclResult_t r) {
    std::snprintf(gc::tls_error, sizeof gc::tls_error, "%s: %s (%d)", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "?", (int)r);
    return GC_E_HIP;
}

#define GC_NCCL(expr)                                     \
    do {                                                  \
        ncclResult_t r__ = (expr);                        \
        if (r__ != ncclSuccess) return rccl_fail(#expr, r__); \
    } while (0)

}  // namespace

struct gc_comm {
    gc_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
    double *d_scratch = nullptr;  // 2 doubles on the device (barrier token / max-reduction)
};

extern "C" {

int gc_comm_available(void) { return rccl().ok ? 1 : 0; }

int gc_comm_version(void) {
    int v = 0;
    if (!rccl().ok || rccl().GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

int gc_comm_get_unique_id(uint8_t *id, size_t len) {
    if (!id || len < GC_COMM_ID_BYTES) return GC_E_ARG;
    if (!rccl().ok) return rccl_missing();
    static_assert(sizeof(ncclUniqueId) == GC_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    GC_NCCL(rccl().GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return GC_OK;
}

static gc_comm *comm_new(gc_ctx *ctx, int *rc) {
    gc_comm *c = new (std::nothrow) gc_comm;
    if (!c) {
        *rc = GC_E_NOMEM;
        return nullptr;
    }
    c->ctx = ctx;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_scratch, 2 * sizeof(double));
    if (e != hipSuccess) {
        gc::set_error("gc_comm", e);
        *rc = GC_E_HIP;
        delete c;
        return nullptr;
    }
    return c;
}

gc_comm *gc_comm_init_rank(gc_ctx *ctx, const uint8_t *id, size_t idlen, int nranks, int rank, int *status) try {
    int rc = GC_OK;
    gc_comm *c = nullptr;
    if (!ctx || !id || idlen < GC_COMM_ID_BYTES || nranks < 1 || rank < 0 || rank >= nranks) rc = GC_E_ARG;
    if (rc == GC_OK && !rccl().ok) rc = rccl_missing();
    if (rc == GC_OK) c = comm_new(ctx, &rc);
    if (rc == GC_OK) {
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof u);
        ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, u, rank);
        if (r != ncclSuccess) {
            rc = rccl_fail("ncclCommInitRank", r);
            c->comm = nullptr;
        }
        c->nranks = nranks;
        c->rank = rank;
    }
    if (rc != GC_OK && c) {
        gc_comm_destroy(c);
        c = nullptr;
    }
    if (status) *status = rc;
    return c;
} catch (...) {
    const int rc__ = gc::on_exception();
    if (status) *status = rc__;
    return nullptr;
}

int gc_comm_init_all(gc_ctx *const *ctxs, int n, gc_comm **out) try {
    if (!ctxs || !out || n < 1) return GC_E_ARG;
    for (int i = 0; i < n; i++) out[i] = nullptr;
    if (!rccl().ok) return rccl_missing();
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return GC_E_ARG;
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; j++)
            if (devs[j] == devs[i]) return GC_E_ARG;  // one rank per device
    }
    std::vector<ncclComm_t> comms(n, nullptr);
    GC_NCCL(rccl().CommInitAll(comms.data(), n, devs.data()));
    int rc = GC_OK;
    for (int i = 0; i < n && rc == GC_OK; i++) {
        out[i] = comm_new(ctxs[i], &rc);
        if (out[i]) {
            out[i]->comm = comms[i];
            out[i]->nranks = n;
            out[i]->rank = i;
            comms[i] = nullptr;
        }
    }
    if (rc != GC_OK) {
        for (int i = 0; i < n; i++) {
            if (comms[i]) (void)rccl().CommDestroy(comms[i]);
            gc_comm_destroy(out[i]);
            out[i] = nullptr;
        }
    }
    return rc;
} catch (...) {
    return gc::on_exception();
}

void gc_comm_destroy(gc_comm *c) {
    if (!c) return;
    if (c->ctx) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
    }
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    delete c;
}

int gc_comm_rank(const gc_comm *c) { return c ? c->rank : -1; }
int gc_comm_nranks(const gc_comm *c) { return c ? c->nranks : 0; }

int gc_comm_allgather(gc_comm *c, const void *d_send, void *d_recv, size_t bytes) {
    if (!c || !c->comm || (bytes && (!d_send || !d_recv))) return GC_E_ARG;
    if (bytes == 0) return GC_OK;
    std::lock_guard<std::mutex> lk(c->ctx->mu);  // the ctx stream is shared with the host-buffer calls
    GC_HIP(hipSetDevice(c->ctx->device));
    GC_NCCL(rccl().AllGather(d_send, d_recv, bytes, ncclUint8, c->comm, c->ctx->stream));
    return GC_OK;
}

int gc_comm_allgather_all(gc_comm *const *cs, int n, const void *const *d_send, void *const *d_recv, size_t bytes) {
    if (!cs || n < 1 || !d_send || !d_recv) return GC_E_ARG;
    for (int i = 0; i < n; i++)
        if (!cs[i] || !cs[i]->comm || cs[i]->nranks != n || (bytes && (!d_send[i] || !d_recv[i]))) return GC_E_ARG;
    if (bytes == 0) return GC_OK;
    GC_NCCL(rccl().GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < n; i++) {
        std::lock_guard<std::mutex> lk(cs[i]->ctx->mu);
        hipError_t e = hipSetDevice(cs[i]->ctx->device);
        ncclResult_t r = e == hipSuccess
                             ? rccl().AllGather(d_send[i], d_recv[i], bytes, ncclUint8, cs[i]->comm, cs[i]->ctx->stream)
                             : ncclUnhandledCudaError;
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    ncclResult_t ge = rccl().GroupEnd();
    if (first != ncclSuccess) return rccl_fail("ncclAllGather", first);
    if (ge != ncclSuccess) return rccl_fail("ncclGroupEnd", ge);
    return GC_OK;
}

// every rank's *value -> the maximum over the ranks (bench timing: max-over-ranks step time); synchronous
int gc_comm_allreduce_max(gc_comm *c, double *value) {
    if (!c || !c->comm || !value) return GC_E_ARG;
    std::lock_guard<std::mutex> lk(c->ctx->mu);
    GC_HIP(hipSetDevice(c->ctx->device));
    GC_HIP(hipMemcpyAsync(c->d_scratch, value, sizeof(double), hipMemcpyHostToDevice, c->ctx->stream));
    GC_NCCL(rccl().AllReduce(c->d_scratch, c->d_scratch + 1, 1, ncclDouble, ncclMax, c->comm, c->ctx->stream));
    GC_HIP(hipMemcpyAsync(value, c->d_scratch + 1, sizeof(double), hipMemcpyDeviceToHost, c->ctx->stream));
    GC_HIP(hipStreamSynchronize(c->ctx->stream));
    return GC_OK;
}

// all ranks have finished everything enqueued on their ctx streams before anybody returns
int gc_comm_barrier(gc_comm *c) {
    double one = 1.0;
    return gc_comm_allreduce_max(c, &one);
}

}  // extern "C"
