// engine.h — internal structs behind the opaque C-ABI handles.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/gcengine.h"
#include "aes_host.h"
#include "kernels.h"
#include "plan.h"

namespace gc {
extern thread_local char tls_error[512];
void set_error(const char *what, hipError_t e);
}  // namespace gc

#define GC_HIP(expr)                                                   \
    do {                                                               \
        hipError_t e__ = (expr);                                       \
        if (e__ != hipSuccess) {                                       \
            gc::set_error(#expr, e__);                                 \
            return e__ == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP; \
        }                                                              \
    } while (0)

struct gc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t *d_te0 = nullptr;  // Te0 (1 KiB), L2-resident source of the LDS tables
    std::mutex mu;              // serialises host-buffer calls sharing this ctx's stream
    bool capturing = false;     // between gc_ctx_capture_begin / _end: launches are recorded, not run
    // host-buffer calls on PINNED caller memory (gc_host_alloc / gc_host_register): the DMA runs on its own stream,
    // chunk k in flight while the transpose kernel of chunk k + 1 fills the other staging buffer (lazily created)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_k[2] = {nullptr, nullptr};  // kernel on stage[b] done
    hipEvent_t ev_c[2] = {nullptr, nullptr};  // copy from / to stage[b] done
    void *stage[2] = {nullptr, nullptr};
    size_t stage_cap = 0;
    // ONE instance of a wide circuit as one cooperative launch (fused_kernels.hip: k_garble_coop): barrier state in device
    // memory, the verdict of the self-test (0 not run, 1 passed, -1 off), and a pinned word a pass raises when a workgroup
    // gives up at a barrier (checked when the next pass is enqueued and by gc_ctx_sync)
    gc::CoopCtl *d_coop = nullptr;
    int coop_state = 0;
    uint32_t *h_coop_err = nullptr;
    uint64_t coop_timeouts = 0;  // passes that lost a workgroup (each was done again on the device: gc_ctx_coop_check)
    uint32_t coop_launched = 0;  // cooperative passes launched on d_coop (= its `passes` counter once they have run)
    // Streaming engine (stream_engine.cpp), kept per ctx so that a stream created per connection does not pay for them again:
    //  * the deep lanes — extra HIP streams that were probed to run beside `stream` (DeepLanes::setup; lanes_state 0: not set
    //    up, 1: ready, -1: none) and the candidates that were set aside;
    //  * the buffers of the launch slots (upload / arena / download regions, device and pinned): taken from and returned to
    //    these lists in power-of-two sizes instead of hipMalloc / hipFree — hipFree waits for the whole device, and a slot
    //    that grows in the middle of a stream would stall every queue.
    int lanes_state = 0;
    std::vector<hipStream_t> lanes, lanes_aside;
    //  * the merged plans of fused step chains and the registry of circuit contents they are keyed by (stream_fuse.cpp)
    struct gcs_fuse_cache *fuse = nullptr;
    std::mutex fuse_mu;
    struct CachedBuf {
        void *p;
        size_t cap;
    };
    std::mutex cache_mu;
    std::vector<CachedBuf> dev_cache, pin_cache;
    size_t dev_cached = 0, pin_cached = 0;
};
void gcs_fuse_cache_free(gc_ctx *);  // stream_fuse.cpp (gc_ctx_destroy: the ctx's merged plans and their planner thread)
namespace gc {
// a buffer of at least `need` bytes (device, or pinned host memory) from the ctx's lists or the runtime; *cap = its real size
hipError_t ctx_buf_get(gc_ctx *c, bool pinned, size_t need, void **p, size_t *cap);
// back to the lists (nothing on the GPU may still use it); freed when the lists hold more than a few GiB
void ctx_buf_put(gc_ctx *c, bool pinned, void *p, size_t cap);
}  // namespace gc
// internal: notes a cooperative pass of this ctx that lost a workgroup (repeated on the device already) and switches the
// cooperative passes off; GC_OK always
int gc_ctx_coop_check(gc_ctx *c);
// internal: device address of the ctx's pinned error word (allocated at first use); nullptr if there is none
uint32_t *gc_ctx_err_word(gc_ctx *c);

struct gc_graph {
    gc_ctx *ctx = nullptr;
    hipGraphExec_t exec = nullptr;
};

struct gc_circ {
    gc_ctx *ctx = nullptr;
    gc_plan plan;
    gc::GateDesc *d_descs = nullptr;
    uint32_t *d_out_slots = nullptr;
    uint32_t *d_slot_of_wire = nullptr;
    gc::Step *d_steps = nullptr;
    uint8_t *d_ops = nullptr;            // op of every gate, original order (table egress / ingest)
    uint32_t *d_row_of_gate = nullptr;   // first slab row of every gate, original order
    uint32_t *d_gwires = nullptr;        // streaming garbler: {in0, in1, out} of every gate, original order (lazy)
    gc::FDesc *d_fdescs = nullptr;   // LDS schedule
    uint32_t *d_fgslot = nullptr;
    gc::Step *d_fsteps = nullptr;
    gc::Chunk *d_fchunks = nullptr;
    uint16_t *d_in_lds = nullptr;
    uint32_t *d_fl_prog = nullptr;   // flattened schedule (Plan::fl_*)
    gc::FUnit *d_fl_units = nullptr;
    uint32_t *d_fl_hgslot = nullptr, *d_fl_ogslot = nullptr;
    uint16_t *d_fl_in_lds = nullptr;
    std::mutex flat_mu;       // the flattened plan and its device arrays are built on first demand (circ_ensure_flat)
    std::atomic<bool> flat_ready{false};
    // ... or, for a streamed circuit seen for the first time, by a thread of the circuit's own while its first passes take the
    // kernels that need no LDS plan (gc_circ_flat_poll): 0 never asked, 1 that thread was started
    std::atomic<int> flat_async{0};
    std::thread flat_thread;
    int schedule = 1;  // default schedule of pooled batches
    bool single_phase = false;
    std::mutex pool_mu;
    std::vector<gc_batch *> pool;  // idle batches reused by gc_garble / gc_eval (cf. garble.go:195-225)
};

struct gc_graph_entry {
    bool eval;
    int rounds;
    int schedule;
    const uint4 *T;
    hipGraphExec_t exec;
};

struct gc_batch {
    gc_circ *circ = nullptr;
    gc::BatchGeom g{};
    uint4 *d_W = nullptr;  // wire labels [nslots][bstride]: L0 (garbler) or active label (evaluator)
    uint4 *d_T = nullptr;  // garbled tables [slab_rows][bstride]
    uint4 *d_R = nullptr;  // [bstride]
    uint32_t *d_rk = nullptr;
    uint64_t *d_prof = nullptr;  // debug cycle breakdown of the fused kernels (gc_batch_debug_profile)
    uint32_t rk_host[60] = {0};
    const gc::StoreXchg *xchg = nullptr;  // gc_pass_dev: (device address) the cooperative pass exchanges labels with the wire store itself
    gc_label r_host{};     // one-instance passes with R from the host: what d_R[0] holds, when r_known
    bool r_known = false;
    int rounds = 0;
    int schedule = 1;
    bool use_graph = true;
    bool store_all = false;      // fused-LDS passes also write every wire to the global array
    bool single_phase = false;   // fused-LDS passes walk the XOR levels (fused_lds_kernels.hip) instead of the flat plan
    bool have_all_wires = false; // the global wire array holds every wire of the last pass
    std::vector<gc_graph_entry> graphs;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    uint32_t last_launches = 0;
};

// internal: gc_circ_load for a CHAIN of streamed circuits fused into one gate list (stream_fuse.cpp): the hash tweak starts
// over at every gate index of seg_first[0 .. nseg) (plan.h: build_plan)
gc_circ *gc_circ_load_seg(gc_ctx *ctx, const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                          uint32_t noutputs, const uint32_t *seg_first, uint32_t nseg, int *status);
// internal (C++ linkage): one instance garbled from explicit labels, the pooled batch kept by the caller
// (tables in b->d_T) until gc_circ_release_batch — used by the streaming garbler's device-side serialiser
int gc_garble_labels_keep(gc_circ *c, const uint8_t *key, size_t keylen, const gc_label *r, const gc_label *inputs,
                          gc_label *out_l0, gc_batch **bout);
void gc_circ_release_batch(gc_circ *c, gc_batch *b);
// internal: ONE instance whose input labels come from, and whose output labels go to, a DEVICE-resident wire store
// (the streaming garbler / evaluator): W[i] = d_store[d_in_idx[i]], pass, d_store[d_out_idx[j]] = W[out_slot[j]]
// (index 0xffffffff: not stored).  eval: the tables come from the host slab (slab_rows labels).  Everything is
// enqueued on the ctx stream; nothing is waited for.  The pooled batch comes back in *bout (tables in b->d_T for the
// garbler's serialiser; later passes on the same stream may reuse it at once).
// internal: the circuit-constant part of a FlatJob for ONE instance of this circuit as one workgroup (ti_log2 = 0) —
// plan arrays, geometry, Te0 — and the dynamic LDS the job needs; false when the circuit has no flattened plan or its
// live labels do not fit next to the AES table.  Builds the flattened plan on first demand (thread-safe).
bool gc_circ_flat_job(gc_circ *c, gc::FlatJob *job, size_t *lds_bytes, bool *has_or);
// internal: is the flattened plan of the circuit built (and on the device)?  If not and `start`, a thread of the circuit's own
// builds it (once); until it is there the circuit's passes run on the kernels that keep the wires in HBM and need no LDS plan
// (a 256-bit multiplier plans for 0.13 s: the streaming evaluator, which meets its circuits in the middle of a stream, does not
// wait for that)
bool gc_circ_flat_poll(gc_circ *c, bool start);
int gc_pass_dev(gc_circ *c, bool eval, const uint8_t *key, size_t keylen, const gc_label *r, const void *d_store,
                const uint32_t *d_in_idx, const uint32_t *d_out_idx, const gc_label *slab_host, size_t slab_rows,
                gc_batch **bout, const gc::StoreXchg *d_xchg = nullptr);
// d_xchg: optional DEVICE copy of gc_pass_xchg(...) — a wide circuit's cooperative pass then exchanges the labels with the
// store itself instead of a kernel before and one behind it
inline gc::StoreXchg gc_pass_xchg(const gc_circ *c, const void *d_store, const uint32_t *d_in_idx, const uint32_t *d_out_idx) {
    return gc::StoreXchg{(uint4 *)const_cast<void *>(d_store), d_in_idx, c->d_out_slots, d_out_idx, c->plan.p.info.noutputs};
}
// internal: a pooled one-instance batch taken BEFORE the pass, so that the caller can upload the tables into b->d_T on a
// stream of its own (under the kernels of the previous pass) and hand the batch to gc_pass_dev in *bout with a null
// slab_host.  A batch held by the caller is not handed out again: two passes in flight get two table buffers.
int gc_pass_batch(gc_circ *c, gc_batch **bout);
