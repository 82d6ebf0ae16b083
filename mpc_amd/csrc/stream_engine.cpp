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
//
// Step-level parallelism (round 3).  A compiled program is a long sequence of SMALL circuits — one per SSA instruction
// (compiler/ssa/streamer.go:412-524: 64-bit adders, comparators, multipliers ...) — most of which do not depend on
// their immediate predecessors.  gc_stream_garble_begin QUEUES such a step; consecutive queued steps that share no
// global wire (no read-after-write, write-after-write or write-after-read through in[] / out[]) form a GROUP that runs
// as ONE launch sequence: host -> device copy of the group's job records, k_*_flat_jobs (workgroup j = step j: its own
// circuit plan, input labels gathered from the device-resident wire store), k_stream_finish (output labels scattered
// into the store, gates serialised into the step's byte slot), one copy of all the group's bytes back into pinned
// memory.  A step that conflicts with the open group closes it (stream order then carries the dependency); the bytes
// still leave in program order through gc_stream_garble_finish.  Steps of more than kSmallGates gates keep the
// per-step path (level launches spread over the chip).  The evaluator groups its blocks the same way.
//
// Deep lanes (round 4).  A step whose one-workgroup pass is LONG (a 128- / 256-bit multiplier, a 256- / 512-bit adder: 0.5 -
// 2 ms on one CU) runs on one of a few extra HIP streams of the ctx, beside the groups, ordered against them by events only
// where two steps share a wire (DeepLanes below); short steps that depend on such a step follow it onto its lane.  The open
// groups form a window of up to 16 (an add chain interleaved with independent products needs a group per link); the caller's
// finish launches what is behind the group it waits for.  A circuit met in the middle of a stream gets its LDS plan from a
// thread of its own while its first passes run on the kernels that need none (gc_circ_flat_poll).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <new>
#include <thread>
#include <unordered_map>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "engine.h"
#include "skel_match.h"

using namespace gc;

namespace {
// developer aid: GC_TRACE=1 prints the wall-clock laps of a streaming step to stderr
struct StreamTrace {
    static bool enabled() {  // (asked once: two getenv walks per streamed step were measurable on 512-gate steps)
        static const bool v = std::getenv("GC_TRACE") != nullptr;
        return v;
    }
    bool on;
    std::chrono::steady_clock::time_point last;
    StreamTrace() : on(enabled()) {
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
    // step groups: can ONE workgroup run this circuit from LDS (-1: not asked yet), and if so the circuit-constant part
    // of its job record
    int small = -1;
    int deep = -1;          // ... and is its pass long enough for a lane of its own (DeepLanes; -1: not asked yet)
    gc::FlatJob job{};
    size_t lds = 0;
    bool has_or = false;
    uint32_t ser_long = 0;  // bytes of the serialised gates with every id in the 4-byte form (an upper bound)
    uint64_t last_use = 0;  // LRU stamp (cache eviction)
    bool pinned = false;    // interned by the caller (gc_stream_intern): never evicted
    size_t cost = 0;        // gates held (host copy + device plan): what the cache budget counts
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
    hipEvent_t up_ev = nullptr;  // behind the latest upload of host-set labels (on the ctx stream): a deep step on a lane of
                                 // its own waits for it (DeepLanes); null until the first upload

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
            // the store moves: nothing may still be writing the old array (deep steps run on lanes of their own, DeepLanes)
            if (d) GC_HIP(hipDeviceSynchronize());
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
        if (!dirty.empty()) {
            if (!up_ev) GC_HIP(hipEventCreateWithFlags(&up_ev, hipEventDisableTiming));
            GC_HIP(hipEventRecord(up_ev, st));
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
        if (up_ev) (void)hipEventDestroy(up_ev);
        up_ev = nullptr;
        d = nullptr;
        cap = 0;
    }
};

// ---- step groups: staging shared by the garbler and the evaluator ----------------------------------------------------
constexpr uint32_t kSmallGates = 32768;          // a step of at most this many gates may join a group (one workgroup, LDS plan)
constexpr uint32_t kSmallWideGates = 8192;       // ... unless it is WIDE (levels of >= 2.5 passes) and larger than this: level
                                                 // launches spread such a circuit over the chip, one workgroup would crawl
constexpr uint32_t kGroupJobs = 256;             // steps per group at most: one workgroup each = one wave of the chip's 256 CUs
constexpr size_t kGroupBytes = (size_t)96 << 20; // wire / table arrays + bytes of one group at most
constexpr uint32_t kMaxPending = 4096;           // circuits queued and not yet finished, at most
constexpr size_t kCacheGatesDefault = (size_t)8 << 20;  // gates the per-stream circuit cache may hold (LRU beyond it)

static inline size_t up16(size_t v) { return (v + 15u) & ~(size_t)15u; }
static inline size_t up256(size_t v) { return (v + 255u) & ~(size_t)255u; }

// one queued small step, host side
struct JobRec {
    CircEntry *ent = nullptr;
    uint32_t nin = 0, nout = 0, ngates = 0, first_tmp = 0, first_out = 0;
    size_t off_io = 0;            // upload region: in[nin], out[nout], out-or-skip[nout] (u32 each)
    size_t off_rows = 0;          // evaluator: the block's table rows in the upload region
    size_t off_w = 0, off_t = 0;  // arena: wire array [nslots], table array [rows]
    size_t off_bytes = 0;         // download region: where the serialised gates go (garbler)
    // evaluator, a deep block of many gates: its rows went up from where the parser put them (pinned ring), on the upload
    // stream, into the slot's arena at off_t (no pass through h_up; the slot's rows_ev says when they are there)
    bool rows_in_arena = false;
};

// the deep steps in flight (DeepLanes) that a step — or a group of steps — has to follow
struct DeepDeps {
    uint32_t ids[6] = {};  // exactly these: the latest deep writers of wires it reads or writes
    uint32_t n = 0;
    uint32_t upto = 0;     // ... and every deep step up to this id: the deep READERS of a wire it writes (the per-wire record
                           // names only the latest of them), or writers beyond what ids[] holds
    bool any() const { return n != 0 || upto != 0; }
    void add(uint32_t id) {
        for (uint32_t i = 0; i < n; i++)
            if (ids[i] == id) return;
        if (n < 6) ids[n++] = id;
        else upto = std::max(upto, id);
    }
    void merge(const DeepDeps &o) {
        for (uint32_t i = 0; i < o.n; i++) add(o.ids[i]);
        upto = std::max(upto, o.upto);
    }
};

struct Slot {
    enum Kind { kFree, kGroup, kBig } kind = kFree;
    gc_ctx *ctx = nullptr;   // the upload / arena / download regions come from (and go back to) the ctx's buffer lists
    bool launched = false, synced = false;
    uint64_t launch_no = 0;  // order of the launches (the evaluator waits for its OLDEST group when it runs out of slots)
    // deep lanes: deep_id != 0: the slot's one job runs on lane `lane` (DeepLanes); deps: the deep steps in flight that the
    // slot's steps have a dependency on (its kernel waits for them)
    uint32_t deep_id = 0;
    DeepDeps deps;
    int lane = -1;
    hipEvent_t dep = nullptr;   // deep: "everything launched on the ctx stream before this step" (the lane waits for it) ...
    hipEvent_t rows_ev = nullptr;   // evaluator, deep: the block's rows are in the arena (JobRec::rows_in_arena; not owned)
    bool after_tail = false;    // ... when the step must follow a pass of the ctx stream that has no event of its own;
    hipEvent_t after_ev = nullptr;  // else the kernel of the latest group it conflicts with (null: none, or done already)
    int error = GC_OK;          // close failed: the group's steps report it
    uint32_t handed = 0;        // steps whose bytes have been handed out
    hipEvent_t kdone = nullptr, done = nullptr;  // kernels of the group enqueued-and-done / bytes back in pinned memory
    hipEvent_t kernel_ev = nullptr;              // whichever of the two says "the group's kernel has run" (set at launch)
    std::vector<JobRec> jobs;
    size_t up_used = 0, arena_used = 0, down_used = 0, lds = 0;
    bool has_or = false;
    uint8_t *h_up = nullptr, *d_up = nullptr, *d_arena = nullptr, *d_down = nullptr, *h_down = nullptr, *d_lane_boff = nullptr;
    size_t h_up_cap = 0, d_up_cap = 0, arena_cap = 0, d_down_cap = 0, h_down_cap = 0, lane_boff_cap = 0;
    // a big step (more than kSmallGates gates): its own wire maps, block offsets, byte buffer and size word
    uint32_t *h_io = nullptr, *d_io = nullptr;  // h_io pinned: the upload is a true asynchronous copy
    size_t h_io_cap = 0, io_cap = 0;
    uint64_t *d_boff = nullptr;
    size_t boff_cap = 0;
    uint8_t *d_bytes = nullptr;
    size_t bytes_cap = 0;
    uint64_t *need = nullptr;   // pinned

    void reset() {
        kind = kFree;
        launched = synced = false;
        error = GC_OK;
        handed = 0;
        deep_id = 0;
        deps = DeepDeps{};
        after_tail = false;
        after_ev = nullptr;
        rows_ev = nullptr;
        lane = -1;
        jobs.clear();
        up_used = arena_used = down_used = lds = 0;
        has_or = false;
    }
    void release() {
        if (ctx) {
            gc::ctx_buf_put(ctx, true, h_up, h_up_cap);
            gc::ctx_buf_put(ctx, false, d_up, d_up_cap);
            gc::ctx_buf_put(ctx, false, d_arena, arena_cap);
            gc::ctx_buf_put(ctx, false, d_down, d_down_cap);
            gc::ctx_buf_put(ctx, true, h_down, h_down_cap);
            gc::ctx_buf_put(ctx, false, d_lane_boff, lane_boff_cap);
        }
        if (h_io) (void)hipHostFree(h_io);
        if (d_io) (void)hipFree(d_io);
        if (d_boff) (void)hipFree(d_boff);
        if (d_bytes) (void)hipFree(d_bytes);
        if (need) (void)hipHostFree(need);
        if (kdone) (void)hipEventDestroy(kdone);
        if (done) (void)hipEventDestroy(done);
        if (dep) (void)hipEventDestroy(dep);
    }
    // pinned upload region with room for `more` further bytes (contents preserved)
    hipError_t reserve_up(size_t more) {
        const size_t need_cap = up_used + more;
        if (need_cap <= h_up_cap) return hipSuccess;
        // (64 KiB to start with: a driver that queues a long chain of dependent small steps gets one slot per step — up to
        // kMaxPending of them —, and a MiB of pinned memory each was gigabytes)
        size_t ncap = std::max<size_t>((size_t)64 << 10, h_up_cap * 2);
        while (ncap < need_cap) ncap *= 2;
        void *n = nullptr;
        hipError_t e = gc::ctx_buf_get(ctx, true, ncap, &n, &ncap);
        if (e != hipSuccess) return e;
        if (up_used) std::memcpy(n, h_up, up_used);
        gc::ctx_buf_put(ctx, true, h_up, h_up_cap);
        h_up = (uint8_t *)n;
        h_up_cap = ncap;
        return hipSuccess;
    }
};

// a slot's region of at least `need` bytes (contents not kept); from the ctx's buffer lists: no hipFree (it would wait for
// every queue of the device) and, after the first stream of a ctx, no hipMalloc either
static hipError_t grow_buf(gc_ctx *ctx, bool pinned, uint8_t **p, size_t *cap, size_t need) {
    if (need <= *cap) return hipSuccess;
    gc::ctx_buf_put(ctx, pinned, *p, *cap);
    *p = nullptr;
    *cap = 0;
    void *n = nullptr;
    size_t ncap = 0;
    hipError_t e = gc::ctx_buf_get(ctx, pinned, need + need / 4, &n, &ncap);
    if (e == hipSuccess) {
        *p = (uint8_t *)n;
        *cap = ncap;
    }
    return e;
}
static hipError_t grow_dev(gc_ctx *ctx, uint8_t **p, size_t *cap, size_t need) { return grow_buf(ctx, false, p, cap, need); }
static hipError_t grow_pin(gc_ctx *ctx, uint8_t **p, size_t *cap, size_t need) { return grow_buf(ctx, true, p, cap, need); }

// The open groups of a stream, oldest first: a small window of launch sequences that have not been launched yet, filled
// by list scheduling.  Group i of the window carries sequence number first_seq + i; per global wire the window remembers
// the sequence number of the latest open group that reads / writes it.  A new step (later in program order than
// everything queued) may join group i only if it conflicts with no step of groups i .. last (it then runs before the
// groups behind i, beside the steps of group i): the earliest such group is one past the latest group it has a
// read-after-write, write-after-write or write-after-read relation with.  Groups are launched in sequence order on one
// HIP stream, so "later group" = "later in time".
// How many groups may be open at once.  A chain of dependent steps interleaved with independent ones — an expression like
// f0*g0 + f1*g9 + ... compiles to mul, mul, add, mul, add, ...: every add follows the add before it — needs one open group per
// link of the chain while the independent steps keep joining the first: with 4 open groups the hundred products of an Ed25519
// field multiplication left in groups of 5 (round 4: 16; GC_STREAM_OPEN_GROUPS for experiments).
// The garbler ties the number to how far ahead its caller queues: a group that stays open keeps gathering steps, but the GPU
// only sees it when the window is full or the caller asks for its bytes — with 64 steps in flight and steps that chain in
// fours, sixteen open groups would hold everything the caller allows and the ctx stream would run dry between two calls of
// gc_stream_garble_finish (ssa23, 64 in flight: 1.1e8 gates/s with 4 open groups, 0.7e8 with 16; the Ed25519 program, 1 024 in
// flight: 0.9e8 with 4, 4.4e8 with 16).  in_flight / 16, between 4 and 16.  The evaluator has no caller waiting for
// results: 16.
constexpr uint32_t kOpenGroupsMin = 4, kOpenGroupsMax = 16;
static uint32_t open_groups_env() {
    static const uint32_t v = [] {
        const char *e = std::getenv("GC_STREAM_OPEN_GROUPS");
        const int n = e && *e ? std::atoi(e) : 0;
        return (uint32_t)std::min(std::max(n, 0), 48);
    }();
    return v;
}
static uint32_t open_groups_limit(size_t in_flight) {
    if (open_groups_env()) return open_groups_env();
    return (uint32_t)std::min<size_t>(std::max<size_t>(in_flight / 16, kOpenGroupsMin), kOpenGroupsMax);
}
struct GroupWindow {
    std::deque<uint32_t> open;      // slots of the open groups, oldest first
    uint32_t first_seq = 1;         // sequence number of open.front()
    std::vector<uint32_t> rd, wr;   // per wire: sequence number of the latest group that reads / writes it (stale if < first_seq)
    void ensure(size_t n) {
        if (rd.size() < n) {
            rd.resize(n, 0);
            wr.resize(n, 0);
        }
    }
    // index into `open` of the earliest group the step may join (== open.size(): it needs a new group)
    uint32_t place(const uint32_t *reads, uint32_t nr, const uint32_t *writes, uint32_t nw) const {
        uint32_t lo = first_seq;
        for (uint32_t i = 0; i < nr; i++) {
            const uint32_t w = wr[reads[i]];
            if (w >= lo) lo = w + 1;  // read after write
        }
        for (uint32_t j = 0; j < nw; j++) {
            if (writes[j] == 0xffffffffu) continue;
            const uint32_t w = wr[writes[j]], r = rd[writes[j]];
            if (w >= lo) lo = w + 1;  // write after write
            if (r >= lo) lo = r + 1;  // write after read
        }
        return lo - first_seq;
    }
    void mark(uint32_t index, const uint32_t *reads, uint32_t nr, const uint32_t *writes, uint32_t nw) {
        const uint32_t seq = first_seq + index;
        for (uint32_t i = 0; i < nr; i++)
            if (rd[reads[i]] < seq) rd[reads[i]] = seq;
        for (uint32_t j = 0; j < nw; j++)
            if (writes[j] != 0xffffffffu) wr[writes[j]] = seq;
    }
    // Launched groups, by sequence number (the last 64): a deep step that conflicts with a step of one of them waits for THAT
    // group's kernel on its lane, not for everything the ctx stream holds.  slot 0xffffffff: a pass of the ctx stream that is
    // no group (a big step): the deep step waits for the stream's tail instead.
    struct Launched {
        uint32_t seq = 0, slot = 0;
        uint64_t launch_no = 0;
    };
    Launched ring[64];
    void note(uint32_t seq, uint32_t slot, uint64_t launch_no) { ring[seq & 63u] = Launched{seq, slot, launch_no}; }
    // sequence number of the latest group — open, launched or long gone — with a step that the step with these reads / writes
    // must follow (0: none)
    uint32_t last_conflict(const uint32_t *reads, uint32_t nr, const uint32_t *writes, uint32_t nw) const {
        uint32_t q = 0;
        for (uint32_t i = 0; i < nr; i++) q = std::max(q, wr[reads[i]]);
        for (uint32_t j = 0; j < nw; j++)
            if (writes[j] != 0xffffffffu) q = std::max(q, std::max(wr[writes[j]], rd[writes[j]]));
        return q;
    }
    // a pass of the ctx stream that is no group (the window is empty: everything queued was launched in front of it) takes a
    // sequence number of its own, so that later deep steps see what it reads and writes
    void mark_pass(const uint32_t *reads, uint32_t nr, const uint32_t *writes, uint32_t nw) {
        mark(0, reads, nr, writes, nw);
        note(first_seq, 0xffffffffu, 0);
        first_seq++;
    }
    // the oldest group leaves the window (it is being launched)
    uint32_t pop() {
        const uint32_t slot = open.front();
        open.pop_front();
        if (++first_seq >= 0xfffffff0u && open.empty()) {  // sequence numbers wrap after 4e9 groups: start over
            std::fill(rd.begin(), rd.end(), 0);
            std::fill(wr.begin(), wr.end(), 0);
            first_seq = 1;
        }
        return slot;
    }
};

struct StepRef {
    uint32_t slot, job;
};

// Groups launched on the ctx stream whose kernels may still be running, oldest first.  The stream is HUNGRY while fewer than
// kKeepQueued of them are.  The garbler looks at it when its caller is about to wait for bytes (gc_stream_garble_finish): the
// open groups behind the one it waits for then go to the GPU too, so that the GPU has a group to run and one behind it while
// the host is busy with the bytes; everything younger stays open and keeps gathering steps.
constexpr size_t kKeepQueued = 2;
struct CtxQueue {
    struct E {
        uint32_t slot;
        uint64_t launch_no;
    };
    std::deque<E> q;
    std::chrono::steady_clock::time_point last{};
    void pushed(uint32_t slot, uint64_t launch_no) { q.push_back(E{slot, launch_no}); }
    template <typename Slots>
    bool hungry(const Slots &slots) {
        if (q.size() < kKeepQueued) return true;
        const auto now = std::chrono::steady_clock::now();
        if (now - last < std::chrono::microseconds(8)) return false;  // (an event query costs a microsecond or two)
        last = now;
        while (!q.empty()) {
            const auto &g = *slots[q.front().slot];
            const bool gone = !g.launched || g.launch_no != q.front().launch_no || g.error != GC_OK;  // its slot was given back
            if (!gone && hipEventQuery(g.kernel_ev) != hipSuccess) break;
            q.pop_front();
        }
        (void)hipGetLastError();  // hipErrorNotReady
        return q.size() < kKeepQueued;
    }
};

// ---- deep lanes ---------------------------------------------------------------------------------------------------------
// A step whose one-workgroup plan is LONG (hundreds of dependent hash phases: a 128- / 256-bit multiplier, a 256- / 512-bit
// adder — 0.5 to 2 ms on one CU) would hold up a whole group of short steps if it joined one, and the whole ctx stream if it
// ran there as a pass of its own.  It runs on a LANE instead: one of a few extra HIP streams, as a group of one job (the same
// kernels, launch sequence and serialiser as a group), beside the groups of the ctx stream and the deep steps of the other
// lanes.  Program order is kept by events, and only where two steps share a wire:
//   * a deep step conflicts with EARLIER small steps -> the open groups that hold them are launched first, and the lane waits
//     for an event recorded on the ctx stream at that point (everything launched there so far: groups, big steps, uploads);
//   * a deep step conflicts with earlier DEEP steps  -> per wire the id of the latest deep reader / writer (ids ascend in
//     program order); the lane waits, per other lane, for that lane's latest step with an id up to the conflicting one (a lane
//     runs in order, so that covers every earlier step of the lane — readers that the per-wire record no longer names too);
//   * a LATER small step conflicts with a deep step   -> its group remembers which (Slot::deps) and the ctx stream
//     waits for the lanes' steps up to it before the group's kernel; a later big step waits for every deep step in flight.
// Every wait names work that was enqueued before the waiter, so the streams cannot deadlock.  Whether a lane really runs
// beside the ctx stream is up to the runtime: it multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the
// environment says otherwise; engine.cpp raises the default to 8 when the library is loaded before the runtime starts), and
// two streams on one queue run one after the other.  Lanes are therefore PROBED when they are created (k_lane_probe): a
// candidate whose kernel cannot see a flag raised by a kernel enqueued afterwards on the ctx stream (or on a lane already
// accepted) shares a queue with it and is set aside; without a usable lane deep steps keep the path of the big steps.
namespace {
__global__ void k_lane_probe(uint32_t *flag, uint32_t *seen, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    uint32_t v = 0;
    while (!(v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    *seen = v;
}
__global__ void k_lane_flag(uint32_t *flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}  // namespace

constexpr uint32_t kDeepStepsDefault = 300;  // barriers per pass (hash phases + XOR rounds) from which a step is deep: ~0.4 ms
constexpr uint32_t kDeepLanesDefault = 3;
constexpr uint32_t kDeepMaxGates = 1u << 20;
constexpr uint32_t kDeepInFlight = 96;       // deep steps launched and not known to be done, at most (then the oldest is waited for)

struct DeepLanes {
    struct InFlight {
        uint32_t id;
        hipEvent_t ev;  // the step's kernel has run (its slot's kdone; the slot is not re-used before retire())
    };
    int state = 0;  // 0: not set up, 1: lanes ready, -1: off
    uint32_t min_steps = kDeepStepsDefault;
    bool follow = true;
    std::vector<hipStream_t> lanes;         // the ctx's lanes (owned by the ctx: set up once, shared by its streams)
    std::vector<std::deque<InFlight>> inflight;  // per lane, ids ascending
    std::vector<uint32_t> rd, wr;           // per global wire: id of the latest deep step that reads / writes it
    uint32_t next_id = 1, n_inflight = 0;
    uint64_t n_steps = 0;

    // the ctx's lanes, created and probed on first use (under ctx->mu)
    static void setup_ctx(gc_ctx *ctx) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (ctx->lanes_state != 0) return;
        ctx->lanes_state = -1;
        const char *v = std::getenv("GC_STREAM_DEEP_LANES");
        int want = v && *v ? std::atoi(v) : (int)kDeepLanesDefault;
        if (want <= 0) return;
        want = std::min(want, 8);
        if (hipSetDevice(ctx->device) != hipSuccess) return;
        uint32_t *h = nullptr, *d_probe = nullptr;
        if (hipMalloc((void **)&d_probe, 2 * sizeof(uint32_t)) != hipSuccess ||
            hipHostMalloc((void **)&h, 2 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            if (d_probe) (void)hipFree(d_probe);
            return;
        }
        // does a kernel on `a` run while one on `b`, enqueued EARLIER, is still spinning?  (b's kernel waits up to 0.4 ms
        // for the flag a's kernel raises)
        auto beside = [&](hipStream_t b, hipStream_t a) -> bool {
            h[0] = h[1] = 0;
            if (hipMemcpyAsync(d_probe, h, 8, hipMemcpyHostToDevice, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return false;
            hipLaunchKernelGGL(k_lane_probe, dim3(1), dim3(1), 0, b, d_probe, d_probe + 1, 40000ull);
            hipLaunchKernelGGL(k_lane_flag, dim3(1), dim3(1), 0, a, d_probe);
            if (hipStreamSynchronize(b) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return false;
            if (hipMemcpy(h, d_probe, 8, hipMemcpyDeviceToHost) != hipSuccess) return false;
            return h[1] == 1;
        };
        const bool trace = std::getenv("GC_TRACE") != nullptr;
        for (int tries = 0; (int)ctx->lanes.size() < want && tries < want + 8; tries++) {
            hipStream_t c = nullptr;
            if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
            bool ok = beside(c, ctx->stream);
            for (size_t i = 0; ok && i < ctx->lanes.size(); i++) ok = beside(c, ctx->lanes[i]);
            if (trace) std::fprintf(stderr, "[gc trace] deep lane candidate %d: %s\n", tries, ok ? "runs beside the ctx stream" : "shares a hardware queue: set aside");
            (ok ? ctx->lanes : ctx->lanes_aside).push_back(c);
        }
        (void)hipGetLastError();
        (void)hipHostFree(h);
        (void)hipFree(d_probe);
        if (!ctx->lanes.empty()) ctx->lanes_state = 1;
    }
    void read_env() {  // (at stream creation: the threshold is asked before the lanes are)
        const char *v = std::getenv("GC_STREAM_DEEP_STEPS");
        if (v && *v) min_steps = (uint32_t)std::max(1, std::atoi(v));
        // GC_STREAM_NO_FOLLOW: short steps never follow a deep step onto its lane (they wait for it in a group)
        follow = std::getenv("GC_STREAM_NO_FOLLOW") == nullptr;
    }
    bool setup(gc_ctx *ctx) {
        if (state != 0) return state > 0;
        state = -1;
        setup_ctx(ctx);
        if (ctx->lanes_state <= 0) return false;
        lanes = ctx->lanes;
        inflight.resize(lanes.size());
        state = 1;
        return true;
    }
    void ensure(size_t n) {
        if (rd.size() < n) {
            rd.resize(n, 0);
            wr.resize(n, 0);
        }
    }
    // the id of the next deep step (ascending in program order; after 4e9 of them: the lanes are drained and the per-wire
    // records start over)
    uint32_t new_id() {
        if (next_id >= 0xfffffff0u) {
            drain();
            std::fill(rd.begin(), rd.end(), 0);
            std::fill(wr.begin(), wr.end(), 0);
            next_id = 1;
        }
        return next_id++;
    }
    // every id up to this one is known to be done
    uint32_t floor() const {
        uint32_t f = next_id - 1;
        for (const auto &q : inflight)
            if (!q.empty()) f = std::min(f, q.front().id - 1);
        return f;
    }
    // the deep steps in flight that a step with these reads / writes depends on
    DeepDeps conflicts(const uint32_t *reads, uint32_t nr, const uint32_t *writes, uint32_t nw) const {
        DeepDeps d;
        if (n_inflight == 0 || rd.empty()) return d;
        const uint32_t fl = floor();
        for (uint32_t i = 0; i < nr; i++)
            if (wr[reads[i]] > fl) d.add(wr[reads[i]]);  // read after write
        for (uint32_t j = 0; j < nw; j++) {
            if (writes[j] == 0xffffffffu) continue;
            if (wr[writes[j]] > fl) d.add(wr[writes[j]]);                                      // write after write
            if (rd[writes[j]] > fl && rd[writes[j]] > wr[writes[j]]) d.upto = std::max(d.upto, rd[writes[j]]);  // write after read
        }
        return d;
    }
    void mark(uint32_t id, const uint32_t *reads, uint32_t nr, const uint32_t *writes, uint32_t nw) {
        for (uint32_t i = 0; i < nr; i++) rd[reads[i]] = id;
        for (uint32_t j = 0; j < nw; j++)
            if (writes[j] != 0xffffffffu) wr[writes[j]] = id;
    }
    // steps whose kernels have run leave the lists (cheap: one query per lane head)
    void poll() {
        for (auto &q : inflight)
            while (!q.empty() && hipEventQuery(q.front().ev) == hipSuccess) {
                q.pop_front();
                n_inflight--;
            }
        (void)hipGetLastError();  // hipErrorNotReady of the queries
    }
    // `st` waits for every deep step with an id up to x (per lane: the latest such step); skip: the lane `st` itself is
    hipError_t wait_upto(hipStream_t st, uint32_t x, int skip) {
        for (size_t l = 0; l < inflight.size(); l++) {
            if ((int)l == skip) continue;
            const auto &q = inflight[l];
            hipEvent_t ev = nullptr;
            for (const InFlight &f : q) {
                if (f.id > x) break;
                ev = f.ev;
            }
            if (ev) {
                hipError_t e = hipStreamWaitEvent(st, ev, 0);
                if (e != hipSuccess) return e;
            }
        }
        return hipSuccess;
    }
    hipError_t wait_all(hipStream_t st) { return n_inflight ? wait_upto(st, next_id, -1) : hipSuccess; }
    // `st` waits for the deep steps of d that are still in flight; skip: the lane `st` itself is (-1: none)
    hipError_t wait_deps(hipStream_t st, const DeepDeps &d, int skip) {
        for (uint32_t i = 0; i < d.n; i++)
            for (size_t l = 0; l < inflight.size(); l++) {  // (a few dozen entries in all; ids ascend inside a lane)
                if ((int)l == skip) continue;
                for (const InFlight &f : inflight[l]) {
                    if (f.id > d.ids[i]) break;
                    if (f.id == d.ids[i]) {
                        hipError_t e = hipStreamWaitEvent(st, f.ev, 0);
                        if (e != hipSuccess) return e;
                        break;
                    }
                }
            }
        return d.upto ? wait_upto(st, d.upto, skip) : hipSuccess;
    }
    // the step is known to be done (its slot is about to be re-used): it and everything older on its lane leave the list
    void retire(int lane, uint32_t id) {
        if (lane < 0 || (size_t)lane >= inflight.size()) return;
        auto &q = inflight[(size_t)lane];
        while (!q.empty() && q.front().id <= id) {
            q.pop_front();
            n_inflight--;
        }
    }
    // the lane of the LATEST deep step in flight among d's (-1: d names none that is still in flight)
    int lane_to_follow(const DeepDeps &d) const {
        uint32_t best = 0;
        for (uint32_t i = 0; i < d.n; i++) best = std::max(best, d.ids[i]);
        best = std::max(best, d.upto);
        if (best == 0 || best <= floor()) return -1;
        // (the lane that holds the latest step in flight with an id up to `best`)
        int lane = -1;
        uint32_t found = 0;
        for (size_t l = 0; l < inflight.size(); l++)
            for (const InFlight &f : inflight[l])
                if (f.id <= best && f.id > found) {
                    found = f.id;
                    lane = (int)l;
                }
        return lane;
    }
    // the least busy lane (fewest steps in flight; ties: the one whose last step is the oldest)
    int pick() const {
        int best = 0;
        for (size_t l = 1; l < inflight.size(); l++) {
            const auto &a = inflight[l], &b = inflight[(size_t)best];
            if (a.size() < b.size() || (a.size() == b.size() && !a.empty() && a.back().id < b.back().id)) best = (int)l;
        }
        return best;
    }
    void drain() {
        for (hipStream_t l : lanes) (void)hipStreamSynchronize(l);
        for (auto &q : inflight) q.clear();
        n_inflight = 0;
    }
    void release() { drain(); }
};

struct gc_stream {
    gc_ctx *ctx = nullptr;
    std::vector<uint8_t> key;
    int rounds = 0;
    gc_label r{};
    uint32_t *d_rk = nullptr;  // expanded key (60 words) and R on the device: the step groups' kernels read them
    uint4 *d_R = nullptr;
    DevStore store;               // global wire -> L0 (L1 = L0 ^ R)
    CircCache cache;
    size_t cache_gates = 0, cache_budget = kCacheGatesDefault;
    uint64_t tick = 0;
    std::vector<uint32_t> alias_gen, alias_j;  // in[] / out[] aliasing check: stamp + index in out[] per global wire
    uint32_t gen = 0;
    std::vector<gc_gate> rewritten;            // gate list with aliased reads redirected (rare)
    std::vector<uint32_t> skip_scratch;        // out[] with 0xffffffff where nothing is stored
    // circuits in flight (gc_stream_garble_begin / _finish), oldest first, and the slots that hold them
    std::vector<std::unique_ptr<Slot>> slots;
    std::deque<StepRef> queue;
    GroupWindow win;              // the groups still accepting steps
    CtxQueue ctxq;                // ... and the launched ones the ctx stream has not run yet
    DeepLanes deep;               // long one-workgroup steps run beside the groups, on streams of their own
    std::vector<CircEntry *> handles;  // gc_stream_intern
    hipStream_t copy_stream = nullptr;
    // Big steps: the serialiser (byte sizes, their scan, the total down, the bytes) runs on ser_stream behind the pass
    // (ev_pass) and under the pass of the NEXT step — which therefore gets another table buffer: the batch of the last big
    // step is held out of the pool until the next one has taken its own, and a pass into a table buffer waits for the
    // serialiser that last read it (ser_ev / ser_batch, two in flight).  Not on copy_stream: the copy of step k's bytes
    // to the host (gc_stream_garble_finish) would queue behind the serialiser of step k + 1, that is behind pass k + 1.
    hipStream_t ser_stream = nullptr;
    hipEvent_t ev_pass = nullptr;
    hipEvent_t ser_ev[2] = {nullptr, nullptr};
    gc_batch *ser_batch[2] = {nullptr, nullptr};
    uint32_t ser_turn = 0;
    gc_circ *held_circ = nullptr;
    gc_batch *held = nullptr;
    uint64_t n_groups = 0, n_group_steps = 0, n_big_steps = 0;
    // gc_stream_garble_finish_view: the slot whose pinned bytes the caller is still reading (given back by the next finish),
    // and pinned staging for the bytes of a big step
    uint32_t view_slot = 0xffffffffu;
    uint8_t *view_buf = nullptr;
    size_t view_cap = 0;
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
// every gate to its byte offset
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
        ser_put(buf + pos, g[k], a.row_of_gate[i], T, lt);
        pos += g[k].size;
    }
}

// ---- the serialiser of a step group (garbler): workgroup j = job j ------------------------------------------------
// The job's gates in the wire format (stream_garble.go:391-446), gate order, into the job's byte slot; the byte count
// into *size_out.  Runs on the copy stream behind the group's garbling kernel, beside the NEXT group's garbling (the
// output labels went back into the wire store in the garbling kernel's own epilogue).
struct FinJob {
    SerArgs a;
    const uint4 *T;
    uint8_t *bytes;
    uint32_t *size_out;
};
constexpr uint32_t kFinThreads = 1024;
__global__ __launch_bounds__(kFinThreads) void k_stream_serialise(const FinJob *jobs) {
    const FinJob j = jobs[blockIdx.x];
    __shared__ uint32_t wsum[kFinThreads / 64];
    const uint32_t per = (j.a.ngates + kFinThreads - 1) / kFinThreads;  // consecutive gates per thread: byte order = gate order
    const uint32_t lo = min(threadIdx.x * per, j.a.ngates), hi = min(lo + per, j.a.ngates);
    uint32_t mine = 0;
    for (uint32_t i = lo; i < hi; i++) mine += ser_gate(j.a, i).size;
    uint32_t incl = mine;
    const uint32_t lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (uint32_t w = 0; w < kFinThreads / 64; w++) {
        if (w < (threadIdx.x >> 6)) base += wsum[w];
        total += wsum[w];
    }
    if (threadIdx.x == 0) *j.size_out = total;
    size_t pos = base + (incl - mine);
    const Layout dense{0, 0, 1, 0};  // one instance: table row r is element r
    for (uint32_t i = lo; i < hi; i++) {
        const SerGate q = ser_gate(j.a, i);
        ser_put(j.bytes + pos, q, j.a.row_of_gate[i], j.T, dense);
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


CircEntry *cache_find(CircCache &cache, uint64_t h, const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t nin,
                      uint32_t nout) {
    auto range = cache.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
        CircEntry &e = it->second;
        if (e.gates.size() != ngates || e.nwires != nwires || e.nin != nin || e.nout != nout) continue;
        bool same = true;
        for (uint32_t i = 0; i < ngates && same; i++)
            same = e.gates[i].in0 == gates[i].in0 && e.gates[i].in1 == gates[i].in1 && e.gates[i].out == gates[i].out &&
                   e.gates[i].op == gates[i].op;
        if (same) return &e;
    }
    return nullptr;
}

// the same on the evaluator's packed gate records (16 bytes, no padding: one memcmp)
CircEntry *cache_find_keys(CircCache &cache, uint64_t h, const std::vector<CircKey> &keys, uint32_t nwires, uint32_t nin,
                           uint32_t nout) {
    static_assert(sizeof(CircKey) == 16, "CircKey must be four packed words");
    auto range = cache.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
        CircEntry &e = it->second;
        if (e.gates.size() != keys.size() || e.nwires != nwires || e.nin != nin || e.nout != nout) continue;
        if (std::memcmp(e.gates.data(), keys.data(), keys.size() * sizeof(CircKey)) == 0) return &e;
    }
    return nullptr;
}

CircEntry *cache_put_keys(CircCache &cache, uint64_t h, gc_circ *circ, const std::vector<CircKey> &keys, uint32_t nwires,
                          uint32_t nin, uint32_t nout) {
    CircEntry e;
    e.circ = circ;
    e.nwires = nwires, e.nin = nin, e.nout = nout;
    e.gates = keys;
    e.cost = keys.size() + 1;
    return &cache.emplace(h, std::move(e))->second;
}

CircEntry *cache_put(CircCache &cache, uint64_t h, gc_circ *circ, const gc_gate *gates, uint32_t ngates, uint32_t nwires,
                     uint32_t nin, uint32_t nout) {
    CircEntry e;
    e.circ = circ;
    e.nwires = nwires, e.nin = nin, e.nout = nout;
    e.gates.resize(ngates);
    uint64_t ser = 0;
    for (uint32_t i = 0; i < ngates; i++) {
        e.gates[i] = CircKey{gates[i].in0, gates[i].in1, gates[i].out, gates[i].op};
        const uint32_t op = gates[i].op;
        ser += 1 + 4u * (op == GC_INV ? 2 : 3) + 16u * (op == GC_AND ? 2 : op == GC_OR ? 3 : op == GC_INV ? 1 : 0);
    }
    e.ser_long = (uint32_t)std::min<uint64_t>(ser, 0xffffffffu);
    e.cost = (size_t)ngates + 1;
    return &cache.emplace(h, std::move(e))->second;
}

// Room for a circuit of `cost` gates in a cache of at most `budget`: least recently used entries go first.  The CALLER
// has made sure nothing on the device or in a queue refers to a cached circuit any more (groups closed, streams drained).
// dropped(circ) is told about every circuit that goes (the evaluator forgets the byte skeletons that point at it).
template <typename F>
void cache_make_room(CircCache &cache, size_t *held, size_t budget, size_t cost, F dropped) {
    while (!cache.empty() && *held + cost > budget) {
        auto victim = cache.end();
        for (auto it = cache.begin(); it != cache.end(); ++it)
            if (!it->second.pinned && (victim == cache.end() || it->second.last_use < victim->second.last_use)) victim = it;
        if (victim == cache.end()) break;  // only interned circuits left
        *held -= std::min(*held, victim->second.cost);
        dropped(victim->second.circ);
        gc_circ_free(victim->second.circ);
        cache.erase(victim);
    }
}

size_t cache_budget_from_env() {
    const char *v = std::getenv("GC_STREAM_CACHE_GATES");
    if (v && *v) {
        const long long n = std::atoll(v);
        if (n > 0) return (size_t)n;
    }
    return kCacheGatesDefault;
}

// can ONE workgroup run this cached circuit from LDS (step groups)?  asked once per entry
bool entry_is_small(CircEntry *e) {
    if (e->small < 0) {
        e->small = 0;
        if (e->gates.size() <= kSmallGates && gc_circ_flat_job(e->circ, &e->job, &e->lds, &e->has_or) &&
            (e->gates.size() <= kSmallWideGates || !wide_for_one_instance(e->circ->plan.p, false)))
            e->small = 1;
    }
    return e->small == 1;
}

// does ONE workgroup run this circuit from LDS, and LONG enough for a lane of its own (DeepLanes)?  Either a step that is too
// big for a group but has a one-workgroup plan (a 128- / 256-bit multiplier on the late schedule), or a small one whose pass
// has at least min_steps barriers (a 256- / 512-bit adder).  A wide circuit is asked nothing: its level launches never need
// the flattened plan (building it for a 131 072-gate step costs more than the step).
// in_stream: the circuit is met in the middle of a stream (the evaluator's blocks; a garbler step that was not interned) —
// the plan of a big one (0.13 s of planning for a 256-bit multiplier) is then built by a thread of the circuit's own, and
// until it is there the answer is "not yet": the step takes the path of the big steps, on kernels that need no LDS plan.
bool entry_is_deep(CircEntry *e, uint32_t min_steps, bool in_stream) {
    if (e->deep < 0) {
        const size_t n = e->gates.size();
        if (n > kDeepMaxGates || (n > kSmallWideGates && wide_for_one_instance(e->circ->plan.p, false))) {
            e->deep = 0;
            return false;
        }
        if (in_stream && n > kSmallGates && !gc_circ_flat_poll(e->circ, true)) return false;
        e->deep = 0;
        const bool ok = e->small == 1 || gc_circ_flat_job(e->circ, &e->job, &e->lds, &e->has_or);
        if (ok && (n > kSmallGates || e->circ->plan.p.n_flat_steps >= min_steps)) e->deep = 1;
    }
    return e->deep == 1;
}

// a free slot; big: for a deep step (megabytes of wire / table arrays) — a free slot that has grown that far already is
// preferred for those and avoided for groups of small steps, so that not every slot of a stream ends up with big regions
Slot *slot_new(gc_ctx *ctx, std::vector<std::unique_ptr<Slot>> &slots, uint32_t *index, bool big = false) {
    constexpr size_t kBigArena = (size_t)2 << 20;
    int other = -1;
    for (uint32_t i = 0; i < slots.size(); i++)
        if (slots[i]->kind == Slot::kFree) {
            if ((slots[i]->arena_cap >= kBigArena) == big) {
                *index = i;
                return slots[i].get();
            }
            if (other < 0) other = (int)i;
        }
    if (other >= 0 && (!big || slots.size() >= 64)) {
        *index = (uint32_t)other;
        return slots[(size_t)other].get();
    }
    std::unique_ptr<Slot> sl(new Slot);
    sl->ctx = ctx;
    if (hipEventCreateWithFlags(&sl->kdone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sl->done, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        sl->release();
        return nullptr;
    }
    slots.push_back(std::move(sl));
    *index = (uint32_t)slots.size() - 1;
    return slots.back().get();
}

// Which launch of the ctx stream does a deep step have to follow on its lane?  cs: sequence number of the latest group with a
// step it conflicts with (GroupWindow::last_conflict; the open ones among them have just been launched).
void deep_after(const GroupWindow &win, const std::vector<std::unique_ptr<Slot>> &slots, uint32_t cs, Slot *ng) {
    if (cs == 0) return;
    const GroupWindow::Launched &l = win.ring[cs & 63u];
    if (l.seq != cs || l.slot == 0xffffffffu) {  // launched too long ago to know, or a pass without an event of its own
        ng->after_tail = true;
        return;
    }
    const Slot &g = *slots[l.slot];
    // (a slot that has been given back or re-used meanwhile: that group was done long ago)
    if (g.kind == Slot::kGroup && g.launched && g.launch_no == l.launch_no && g.error == GC_OK) ng->after_ev = g.kernel_ev;
}

// Launch sequence of a group (see the head of this file).  eval: the jobs' table rows are part of the upload region and
// nothing comes back.  On return the slot is `launched`; a failure is kept in slot.error for the group's steps.
// A deep step (g.deep_id != 0, one job) takes the same sequence on its lane, behind an event recorded on the ctx stream here
// and behind the deep steps of the other lanes named by g.deps; a group of small steps on the ctx stream waits for the deep
// steps of g.deps (DeepLanes).
int launch_group(gc_ctx *ctx, Slot &g, bool eval, DevStore &store, const uint32_t *d_rk, const uint4 *d_R, int rounds,
                 hipStream_t copy_stream, DeepLanes &deep) {
    const bool on_lane = g.deep_id != 0;
    hipStream_t st = on_lane ? deep.lanes[(size_t)g.lane] : ctx->stream;
    std::lock_guard<std::mutex> lk(ctx->mu);
    g.launched = true;
    static std::atomic<uint64_t> launches{0};
    g.launch_no = ++launches;
    auto fail = [&](const char *what, hipError_t e) {
        set_error(what, e);
        g.error = e == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
        return g.error;
    };
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail("launch_group", e);
    int rcs = store.flush(ctx);  // host-set labels go up first; the store may move (its pointer is taken below)
    if (rcs != GC_OK) return g.error = rcs;
    if (on_lane) {
        if (store.up_ev) e = hipStreamWaitEvent(st, store.up_ev, 0);  // host-set labels it may read (long done, as a rule)
        if (e == hipSuccess && g.after_tail) {
            if (!g.dep) e = hipEventCreateWithFlags(&g.dep, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(g.dep, ctx->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, g.dep, 0);
        } else if (e == hipSuccess && g.after_ev) {
            e = hipStreamWaitEvent(st, g.after_ev, 0);
        }
        if (e == hipSuccess && g.deps.any()) e = deep.wait_deps(st, g.deps, g.lane);
    } else if (g.deps.any()) {
        deep.poll();
        e = deep.wait_deps(st, g.deps, -1);
    }
    if (e != hipSuccess) return fail("launch_group (order)", e);
    const uint32_t n = (uint32_t)g.jobs.size();
    const size_t off_fj = up16(g.up_used), off_fin = off_fj + (size_t)n * sizeof(FlatJob);
    const size_t total_up = off_fin + (size_t)n * sizeof(FinJob);
    const size_t sizes_bytes = up256((size_t)n * sizeof(uint32_t));
    if ((e = g.reserve_up(total_up - g.up_used)) != hipSuccess) return fail("launch_group (pinned)", e);
    if ((e = grow_dev(ctx, &g.d_up, &g.d_up_cap, total_up)) != hipSuccess) return fail("launch_group (upload)", e);
    if ((e = grow_dev(ctx, &g.d_arena, &g.arena_cap, std::max<size_t>(g.arena_used, 256))) != hipSuccess) return fail("launch_group (arena)", e);
    if (!eval) {
        if ((e = grow_dev(ctx, &g.d_down, &g.d_down_cap, sizes_bytes + g.down_used)) != hipSuccess) return fail("launch_group (bytes)", e);
        if ((e = grow_pin(ctx, &g.h_down, &g.h_down_cap, sizes_bytes + g.down_used)) != hipSuccess) return fail("launch_group (pinned bytes)", e);
    }
    FlatJob *fj = (FlatJob *)(g.h_up + off_fj);
    FinJob *fin = (FinJob *)(g.h_up + off_fin);
    for (uint32_t k = 0; k < n; k++) {
        const JobRec &j = g.jobs[k];
        const uint32_t *d_io = (const uint32_t *)(g.d_up + j.off_io);
        FlatJob f = j.ent->job;
        f.W = (uint4 *)(g.d_arena + j.off_w);
        f.T = eval && !j.rows_in_arena ? (uint4 *)(g.d_up + j.off_rows) : (uint4 *)(g.d_arena + j.off_t);
        f.R = d_R;
        f.Rout = nullptr;
        f.rk = d_rk;
        f.store = store.d;
        f.in_idx = d_io;
        f.out_slots = j.ent->circ->d_out_slots;
        f.out_idx = eval ? d_io + j.nin : d_io + j.nin + j.nout;
        f.nout = j.nout;
        fj[k] = f;
        FinJob q{};
        if (!eval) {
            q.a.gw = j.ent->circ->d_gwires;
            q.a.ops = j.ent->circ->d_ops;
            q.a.row_of_gate = j.ent->circ->d_row_of_gate;
            q.a.in = d_io;
            q.a.out = d_io + j.nin;
            q.a.ngates = j.ngates;
            q.a.first_tmp = j.first_tmp;
            q.a.first_out = j.first_out;
            q.bytes = g.d_down + sizes_bytes + j.off_bytes;
            q.size_out = (uint32_t *)g.d_down + k;
        }
        q.T = f.T;
        fin[k] = q;
    }
    e = hipMemcpyAsync(g.d_up, g.h_up, total_up, hipMemcpyHostToDevice, st);  // pinned source: a true asynchronous copy
    if (e == hipSuccess && g.rows_ev) e = hipStreamWaitEvent(st, g.rows_ev, 0);
    if (e == hipSuccess) e = launch_fused_flat_jobs(eval, rounds, g.has_or, (const FlatJob *)(g.d_up + off_fj), n, g.lds, st);
    // "the group's kernel has run": kdone for the garbler (the serialiser and the bytes' way back follow on the copy stream),
    // done itself for the evaluator (nothing follows)
    g.kernel_ev = eval ? g.done : g.kdone;
    if (e == hipSuccess) e = hipEventRecord(g.kernel_ev, st);
    if (e == hipSuccess && on_lane) {
        deep.inflight[(size_t)g.lane].push_back(DeepLanes::InFlight{g.deep_id, g.kernel_ev});
        deep.n_inflight++;
        deep.n_steps++;
    }
    if (e == hipSuccess && !eval) {
        // serialiser and bytes on the copy stream: the next group's garbling need not wait for either
        e = hipStreamWaitEvent(copy_stream, g.kdone, 0);
        if (e == hipSuccess && on_lane && n == 1 && g.jobs[0].ngates > 4 * kSerGates) {
            // a deep step of many gates: the serialiser of the big steps, spread over the chip (one workgroup would write
            // megabytes byte by byte: 1 - 2 ms on the copy stream, more than the step's pass)
            const FinJob &q = fin[0];
            const uint32_t nblocks = (q.a.ngates + kSerGates - 1) / kSerGates;
            if (g.lane_boff_cap < ((size_t)nblocks + 1) * sizeof(uint64_t))
                e = grow_dev(ctx, &g.d_lane_boff, &g.lane_boff_cap, ((size_t)nblocks + 1) * sizeof(uint64_t));
            uint64_t *boff = (uint64_t *)g.d_lane_boff;
            if (e == hipSuccess) {
                const Layout dense{0, 0, 1, 0};
                hipLaunchKernelGGL(k_ser_sizes, dim3(nblocks), dim3(kSerThreads), 0, copy_stream, q.a, boff);
                hipLaunchKernelGGL(k_ser_scan, dim3(1), dim3(1024), 0, copy_stream, boff, nblocks, q.size_out);
                hipLaunchKernelGGL(k_ser_write, dim3(nblocks), dim3(kSerThreads), 0, copy_stream, q.a, boff, q.T, dense, q.bytes);
                e = hipGetLastError();
            }
        } else if (e == hipSuccess) {
            hipLaunchKernelGGL(k_stream_serialise, dim3(n), dim3(kFinThreads), 0, copy_stream, (const FinJob *)(g.d_up + off_fin));
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(g.h_down, g.d_down, sizes_bytes + g.down_used, hipMemcpyDeviceToHost, copy_stream);
        if (e == hipSuccess) e = hipEventRecord(g.done, copy_stream);
    }
    if (e != hipSuccess) return fail("launch_group", e);
    return GC_OK;
}

// the oldest open group leaves the window and is launched
int launch_oldest(gc_stream *s) {
    if (s->win.open.empty()) return GC_OK;
    const uint32_t seq = s->win.first_seq, slot = s->win.pop();
    Slot &g = *s->slots[slot];
    s->n_groups++;
    s->n_group_steps += g.jobs.size();
    const int rc = launch_group(s->ctx, g, false, s->store, s->d_rk, s->d_R, s->rounds, s->copy_stream, s->deep);
    s->win.note(seq, slot, g.launch_no);
    if (rc == GC_OK) s->ctxq.pushed(slot, g.launch_no);
    return rc;
}
// everything queued is launched, in order (a read-back, a big step or the caller's flush follows)
int close_group(gc_stream *s) {
    int rc = GC_OK;
    while (!s->win.open.empty()) {
        const int r = launch_oldest(s);
        if (rc == GC_OK) rc = r;
    }
    return rc;
}

// the stream's cached device circuit for this gate list (cached by content; a new circuit is validated once:
// garbleGate's checks, stream_garble.go:195-210)
int stream_find_or_load(gc_stream *s, const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t nin, uint32_t nout,
                        CircEntry **out_ent) {
    const uint32_t first_tmp = nin;
    hipStream_t st = s->ctx->stream;
    CircEntry *ent = nullptr;
    {
        const uint64_t h = circuit_hash(gates, ngates, nwires, nin, nout);
        ent = cache_find(s->cache, h, gates, ngates, nwires, nin, nout);
        if (!ent) {
            for (uint32_t i = 0; i < ngates; i++) {
                if (gates[i].op > GC_INV) return GC_E_GATE;
                if (gates[i].out < first_tmp) return GC_E_ARG;  // a gate writing an input-mapped wire: not produced by the compiler
            }
            if (s->cache_gates + ngates + 1 > s->cache_budget && !s->cache.empty()) {
                // over budget: least recently used circuits go.  Nothing may refer to them any more: launch what is queued
                // and drain both streams first (rare: once per budget's worth of NEW circuits)
                int rcq = close_group(s);
                if (rcq != GC_OK) return rcq;
                GC_HIP(hipStreamSynchronize(st));
                GC_HIP(hipStreamSynchronize(s->copy_stream));
                if (s->ser_stream) GC_HIP(hipStreamSynchronize(s->ser_stream));
                s->deep.drain();
                GC_HIP(hipStreamSynchronize(s->copy_stream));  // (the serialisers of the deep steps)
                if (s->held) gc_circ_release_batch(s->held_circ, s->held);  // (back into its circuit's pool before that may go)
                s->held = nullptr;
                s->ser_batch[0] = s->ser_batch[1] = nullptr;
                cache_make_room(s->cache, &s->cache_gates, s->cache_budget, (size_t)ngates + 1, [](gc_circ *) {});
            }
            int stc = GC_OK;
            gc_circ *circ = gc_circ_load(s->ctx, gates, ngates, nwires, nin, nout, &stc);
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
            ent = cache_put(s->cache, h, circ, gates, ngates, nwires, nin, nout);
            s->cache_gates += ent->cost;
        }
    }
    *out_ent = ent;
    return GC_OK;
}

int stream_begin(gc_stream *s, const gc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in, uint32_t nin,
                 const uint32_t *out, uint32_t nout, CircEntry *known);

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
        s->rounds = k.rounds;
        s->cache_budget = cache_budget_from_env();
        s->deep.read_env();
        s->r = gc_label{be64(rnd) | 0x8000000000000000ull, be64(rnd + 8)};  // R.SetS(true)
        uint32_t mx = 0;
        for (uint32_t i = 0; i < ninputs; i++) mx = std::max(mx, inputs[i]);
        ensure(s, mx);
        for (uint32_t i = 0; i < ninputs; i++)
            s->store.set(inputs[i], gc_label{be64(rnd + 16 * ((size_t)i + 1)), be64(rnd + 16 * ((size_t)i + 1) + 8)});
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_rk, sizeof k.w);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_R, sizeof(uint4));
        if (e == hipSuccess) e = hipMemcpy(s->d_rk, k.w, sizeof k.w, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->d_R, &s->r, sizeof(gc_label), hipMemcpyHostToDevice);
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
    s->deep.release();
    if (s->copy_stream) {
        (void)hipStreamSynchronize(s->copy_stream);
        (void)hipStreamDestroy(s->copy_stream);
    }
    if (s->ser_stream) {
        (void)hipStreamSynchronize(s->ser_stream);
        (void)hipStreamDestroy(s->ser_stream);
    }
    for (hipEvent_t ev : {s->ev_pass, s->ser_ev[0], s->ser_ev[1]})
        if (ev) (void)hipEventDestroy(ev);
    if (s->held) gc_circ_release_batch(s->held_circ, s->held);
    for (auto &kv : s->cache) gc_circ_free(kv.second.circ);
    for (auto &sl : s->slots) sl->release();
    if (s->ctx) gc::ctx_buf_put(s->ctx, true, s->view_buf, s->view_cap);
    if (s->d_rk) (void)hipFree(s->d_rk);
    if (s->d_R) (void)hipFree(s->d_R);
    s->store.release();
    delete s;
}

// gc_stream_intern: the circuit is looked up (or loaded) once and named by a handle; gc_stream_garble_begin_h then
// skips the per-call content hash and gate-by-gate comparison — for a 4 096-gate SSA-step circuit that is most of the
// host's share of a step.  Interned circuits are never evicted from the cache.
int gc_stream_intern(gc_stream *s, const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t nin, uint32_t nout,
                     uint32_t *handle) try {
    if (!s || !handle || (!gates && ngates) || ngates == 0 || nin > nwires || nout > nwires) return GC_E_ARG;
    CircEntry *ent = nullptr;
    int rc = stream_find_or_load(s, gates, ngates, nwires, nin, nout, &ent);
    if (rc != GC_OK) return rc;
    ent->pinned = true;
    // interning is where a circuit's one-time work belongs: the flattened plan (built on first demand: 0.1 s for a 256-bit
    // multiplier, early and late schedule) and, for a step that runs as a pass of its own, the two batches it alternates
    // between — not inside the caller's first steps
    if (entry_is_deep(ent, s->deep.min_steps, false) && s->deep.setup(s->ctx)) {
        // (a long one-workgroup pass: it will run on a lane, as a group of one job — the lanes are set up here as well)
    } else if (!entry_is_small(ent)) {
        gc_batch *b0 = nullptr, *b1 = nullptr;
        if (gc_pass_batch(ent->circ, &b0) == GC_OK && gc_pass_batch(ent->circ, &b1) == GC_OK) {
        }
        if (b0) gc_circ_release_batch(ent->circ, b0);
        if (b1) gc_circ_release_batch(ent->circ, b1);
    }
    uint32_t spare = (uint32_t)s->handles.size();
    for (uint32_t i = 0; i < s->handles.size(); i++) {
        if (s->handles[i] == ent) {
            *handle = i;
            return GC_OK;
        }
        if (!s->handles[i] && spare == s->handles.size()) spare = i;  // (a released handle's number is used again)
    }
    if (spare == s->handles.size()) s->handles.push_back(nullptr);
    s->handles[spare] = ent;
    *handle = spare;
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_stream_release(gc_stream *s, uint32_t handle) try {
    if (!s || handle >= s->handles.size() || !s->handles[handle]) return GC_E_ARG;
    s->handles[handle]->pinned = false;  // an ordinary cache entry from here on (steps in flight keep using it: eviction
    s->handles[handle] = nullptr;        // waits for them, stream_find_or_load)
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_stream_garble_begin_h(gc_stream *s, uint32_t handle, const uint32_t *in, const uint32_t *out) try {
    if (!s || handle >= s->handles.size() || !s->handles[handle]) return GC_E_ARG;
    CircEntry *ent = s->handles[handle];
    return stream_begin(s, nullptr, (uint32_t)ent->gates.size(), ent->nwires, in, ent->nin, out, ent->nout, ent);
} catch (...) {
    return gc::on_exception();
}

int gc_stream_garble_flush(gc_stream *s) try {
    if (!s) return GC_E_ARG;
    return close_group(s);
} catch (...) {
    return gc::on_exception();
}

int gc_stream_stats(const gc_stream *s, uint64_t *groups, uint64_t *grouped_steps, uint64_t *big_steps) {
    if (!s) return GC_E_ARG;
    if (groups) *groups = s->n_groups;
    if (grouped_steps) *grouped_steps = s->n_group_steps;
    if (big_steps) *big_steps = s->n_big_steps;
    return GC_OK;
}

int gc_stream_deep_stats(const gc_stream *s, uint64_t *deep_steps, uint32_t *lanes) {
    if (!s) return GC_E_ARG;
    if (deep_steps) *deep_steps = s->deep.n_steps;
    if (lanes) *lanes = s->deep.state > 0 ? (uint32_t)s->deep.lanes.size() : 0;
    return GC_OK;
}

int gc_stream_get_wire(gc_stream *s, uint32_t w, gc_wire *out) try {  // Streaming.GetInput (:117-119)
    if (!s || !out) return GC_E_ARG;
    int rc = close_group(s);  // a queued step may be the one that sets the wire
    if (rc != GC_OK) return rc;
    if (s->deep.n_inflight) s->deep.drain();  // ... or a deep step on its lane
    gc_label l0;
    rc = s->store.get(s->ctx, w, &l0);
    if (rc != GC_OK) return rc;
    (void)gc_ctx_coop_check(s->ctx);  // (the stream was waited for: a cooperative pass that lost a workgroup is noted here)
    out->l0 = l0;
    out->l1 = gc_label{l0.d0 ^ s->r.d0, l0.d1 ^ s->r.d1};
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

// Streaming.Garble in two halves (additive): _begin queues one circuit and returns WITHOUT waiting; _finish hands out the
// bytes of the oldest circuit in flight.  With begin(k + 1 ...) before finish(k) the host's share of a step overlaps the
// GPU's share of the steps before, and small independent steps share one launch sequence (step groups, see the head of
// this file): the bytes still leave in order.  At most kMaxPending circuits in flight.
int gc_stream_garble_begin(gc_stream *s, const gc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                           uint32_t nin, const uint32_t *out, uint32_t nout) try {
    if (!s || (!gates && ngates)) return GC_E_ARG;
    return stream_begin(s, gates, ngates, nwires, in, nin, out, nout, nullptr);
} catch (...) {
    return gc::on_exception();
}

// known: the interned circuit of gc_stream_garble_begin_h (gates == nullptr then)
}  // extern "C"

namespace {
int stream_begin(gc_stream *s, const gc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                 uint32_t nin, const uint32_t *out, uint32_t nout, CircEntry *known) {
    if (!s || (nin && !in) || (nout && !out)) return GC_E_ARG;
    if (s->queue.size() >= kMaxPending) return GC_E_ARG;
    // in[] and out[] may overlap (a circuit whose last wires are input wires): initCircuit (:102-114) takes both as they
    // are, Get / Set resolve a wire through in[] first (:131-157), so such an output id is simply never written
    if (nin > nwires || nout > nwires) return GC_E_ARG;
    const uint32_t first_tmp = nin, first_out = nwires - nout;
    // initCircuit (:102-114)
    uint32_t mx = 0;
    for (uint32_t i = 0; i < nin; i++) mx = std::max(mx, in[i]);
    for (uint32_t i = 0; i < nout; i++) mx = std::max(mx, out[i]);
    ensure(s, mx);
    gc_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    GC_HIP(hipSetDevice(ctx->device));
    StreamTrace tr;

    // in[] / out[] naming the same GLOBAL wire (wire-id re-use, in-place update): the reference resolves
    // stream.wire(index) per gate (:131-157), so a gate that reads the input-mapped wire after the gate that Set the
    // output-mapped one sees the NEW label.  The device garbles from a snapshot of the inputs: redirect such reads to
    // the producing circuit wire (same global id and flags on the wire, so the serialised bytes do not change).
    if (ngates) {
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
            if (known) {  // an interned circuit bound so that an output updates one of its inputs in place: rare, general path
                s->rewritten.assign(ngates, gc_gate{});
                for (uint32_t g = 0; g < ngates; g++) {
                    const CircKey &k = known->gates[g];
                    s->rewritten[g].in0 = k.in0, s->rewritten[g].in1 = k.in1, s->rewritten[g].out = k.out;
                    s->rewritten[g].op = (uint8_t)k.op;
                }
                known = nullptr;
            } else {
                s->rewritten.assign(gates, gates + ngates);
            }
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
    CircEntry *ent = known;
    if (ngates && !ent) {
        int rcl = stream_find_or_load(s, gates, ngates, nwires, nin, nout, &ent);
        if (rcl != GC_OK) return rcl;
    }
    if (ent) ent->last_use = ++s->tick;
    tr.lap("alias + hash + cache");

    // out[] with "no store" marks (an output wire that is an input wire has no gate: no Set)
    s->skip_scratch.resize(nout);
    for (uint32_t j = 0; j < nout; j++) s->skip_scratch[j] = first_out + j >= first_tmp ? out[j] : 0xffffffffu;

    // ---- a small step joins the earliest open group it has no dependency on (or behind); a deep one takes a lane ---------
    bool is_deep = ngates && entry_is_deep(ent, s->deep.min_steps, known == nullptr) && s->deep.setup(ctx);
    // A SHORT step that reads or overwrites what a deep step in flight writes or reads FOLLOWS that step onto its lane (as a
    // deep step of its own: a group of one job, ordered by the lane).  In a group it would make the ctx stream wait for the
    // deep step — with every group behind it, whether they have anything to do with it or not (an in-order stream); on the
    // lane only the chain that really depends on the long step waits for it (ssa23: the 256- and 512-bit values chain among
    // themselves; the ctx stream idled 74 of 286 ms behind multipliers before).
    int follow_lane = -1;
    if (!is_deep && ngates && s->deep.n_inflight && s->deep.follow && entry_is_small(ent)) {
        s->deep.ensure(s->store.host.size());
        follow_lane = s->deep.lane_to_follow(s->deep.conflicts(in, nin, s->skip_scratch.data(), nout));
        is_deep = follow_lane >= 0;
    }
    if (is_deep || (ngates && entry_is_small(ent))) {
        // labels the host has set and not uploaded yet go up BEFORE this step's outputs are marked device-owned (the
        // upload skips device-owned wires: an output that overwrites a host-set input of the same step would lose it) —
        // and before the step is put anywhere: a failure here leaves nothing half-queued
        if (!s->store.dirty.empty()) {
            std::lock_guard<std::mutex> lk(ctx->mu);
            int rcs = s->store.flush(ctx);
            if (rcs != GC_OK) return rcs;
        }
        s->win.ensure(s->store.host.size());
        if (is_deep || s->deep.n_inflight) s->deep.ensure(s->store.host.size());
        const size_t wbytes = up256((size_t)ent->job.w_tile * 16) + up256((size_t)ent->job.t_tile * 16);
        uint32_t gi = s->win.place(in, nin, s->skip_scratch.data(), nout);
        uint32_t slot_idx = 0;
        if (is_deep) {
            // the open groups this step depends on go to the GPU first (place(): every conflict sits in a group before gi)
            for (; gi > 0; gi--) {
                int rcq = launch_oldest(s);
                if (rcq != GC_OK) return rcq;
            }
            s->deep.poll();
            if (s->deep.n_inflight >= kDeepInFlight) {  // bounded: wait for the oldest deep step of the fullest lane
                size_t l = 0;
                for (size_t k = 1; k < s->deep.inflight.size(); k++)
                    if (s->deep.inflight[k].size() > s->deep.inflight[l].size()) l = k;
                (void)hipEventSynchronize(s->deep.inflight[l].front().ev);
                s->deep.poll();
            }
            Slot *ng = slot_new(ctx, s->slots, &slot_idx, follow_lane < 0);
            if (!ng) return GC_E_NOMEM;
            ng->reset();
            ng->kind = Slot::kGroup;
            ng->deep_id = s->deep.new_id();
            ng->lane = follow_lane >= 0 ? follow_lane : s->deep.pick();
            ng->deps = s->deep.conflicts(in, nin, s->skip_scratch.data(), nout);
            deep_after(s->win, s->slots, s->win.last_conflict(in, nin, s->skip_scratch.data(), nout), ng);
        } else {
            auto full = [&](const Slot &g) {
                return g.jobs.size() >= kGroupJobs || g.arena_used + g.down_used + wbytes + ent->ser_long > kGroupBytes;
            };
            while (gi < s->win.open.size() && full(*s->slots[s->win.open[gi]])) gi++;
            if (gi == s->win.open.size()) {  // behind every open group: a new one (the oldest goes to the GPU when the window is full)
                if (s->win.open.size() >= open_groups_limit(s->queue.size())) {
                    int rcq = launch_oldest(s);
                    if (rcq != GC_OK) return rcq;
                    gi--;
                }
                uint32_t idx = 0;
                Slot *ng = slot_new(ctx, s->slots, &idx);
                if (!ng) return GC_E_NOMEM;
                ng->reset();
                ng->kind = Slot::kGroup;
                s->win.open.push_back(idx);
            }
            slot_idx = s->win.open[gi];
        }
        Slot &g = *s->slots[slot_idx];
        const size_t io_bytes = up16(((size_t)nin + 2 * (size_t)nout) * sizeof(uint32_t));
        hipError_t e = g.reserve_up(up16(g.up_used) - g.up_used + io_bytes);
        if (e != hipSuccess) {
            set_error("gc_stream_garble (pinned)", e);
            if (is_deep) g.reset();
            return GC_E_NOMEM;
        }
        JobRec j;
        j.ent = ent;
        j.nin = nin, j.nout = nout, j.ngates = ngates, j.first_tmp = first_tmp, j.first_out = first_out;
        g.up_used = up16(g.up_used);
        j.off_io = g.up_used;
        uint32_t *io = (uint32_t *)(g.h_up + g.up_used);
        if (nin) std::memcpy(io, in, (size_t)nin * sizeof(uint32_t));
        if (nout) {
            std::memcpy(io + nin, out, (size_t)nout * sizeof(uint32_t));
            std::memcpy(io + nin + nout, s->skip_scratch.data(), (size_t)nout * sizeof(uint32_t));
        }
        g.up_used += io_bytes;
        j.off_w = g.arena_used;
        g.arena_used += up256((size_t)ent->job.w_tile * 16);
        j.off_t = g.arena_used;
        g.arena_used += up256((size_t)ent->job.t_tile * 16);
        j.off_bytes = g.down_used;
        g.down_used += up16((size_t)ent->ser_long);
        g.lds = std::max(g.lds, ent->lds);
        g.has_or = g.has_or || ent->has_or;
        g.jobs.push_back(j);
        if (!is_deep) {
            s->win.mark(gi, in, nin, s->skip_scratch.data(), nout);
            // a deep step in flight that this one must follow: the group waits for it (and for the older ones of its lane)
            if (s->deep.n_inflight) g.deps.merge(s->deep.conflicts(in, nin, s->skip_scratch.data(), nout));
        }
        if (is_deep) {  // launched at once, on its lane
            int rcl = launch_group(ctx, g, false, s->store, s->d_rk, s->d_R, s->rounds, s->copy_stream, s->deep);
            if (rcl != GC_OK) {
                (void)hipStreamSynchronize(s->deep.lanes[(size_t)g.lane]);
                s->deep.retire(g.lane, g.deep_id);
                g.reset();
                return rcl;
            }
            s->deep.mark(g.deep_id, in, nin, s->skip_scratch.data(), nout);
            s->n_groups++;
            s->n_group_steps++;
        }
        for (uint32_t k = 0; k < nout; k++)
            if (s->skip_scratch[k] != 0xffffffffu) s->store.on_dev[out[k]] = 1;
        s->queue.push_back(StepRef{slot_idx, (uint32_t)g.jobs.size() - 1});
        tr.lap(is_deep ? "launched on a lane" : "queued in group");
        return GC_OK;
    }

    // ---- a big (or empty) step: its own launch sequence, behind everything queued ------------------------------
    {
        int rcq = close_group(s);
        if (rcq != GC_OK) return rcq;
        if (s->deep.n_inflight) {  // a pass on the ctx stream: behind every deep step in flight (DeepLanes)
            s->deep.poll();
            std::lock_guard<std::mutex> lk(ctx->mu);
            GC_HIP(s->deep.wait_all(st));
        }
        if (ngates) {  // ... and later deep steps must see what it reads and writes
            s->win.ensure(s->store.host.size());
            s->win.mark_pass(in, nin, s->skip_scratch.data(), nout);
        }
    }
    uint32_t idx = 0;
    Slot *bg = slot_new(ctx, s->slots, &idx);
    if (!bg) return GC_E_NOMEM;
    bg->reset();
    Slot &b = *bg;
    if (!b.need) GC_HIP(hipHostMalloc((void **)&b.need, sizeof(uint64_t), hipHostMallocDefault));
    if (ngates == 0) {  // nothing on the wire
        *b.need = 0;
        GC_HIP(hipEventRecord(b.done, st));
        b.kind = Slot::kBig;
        b.launched = true;
        s->queue.push_back(StepRef{idx, 0});
        return GC_OK;
    }
    gc_circ *circ = ent->circ;
    const uint32_t nblocks = (ngates + kSerGates - 1) / kSerGates;
    // the table buffer of this step: not the one of the step before, whose serialiser may still be reading it
    gc_batch *bt = nullptr;
    {
        int rcb = gc_pass_batch(circ, &bt);
        if (rcb != GC_OK) return rcb;
    }
    // (1) this call's wire maps on the device: in[], out[], and out[] with "no store" marks for the scatter; host-set
    //     labels of the store are uploaded.  The maps go up from the slot's own PINNED staging: a true asynchronous copy
    //     that no later call can overwrite (every step in flight has its own slot).
    const size_t io_words = (size_t)nin + 2 * (size_t)nout + 1;
    // ... and behind them the pass's label exchange record (gc_pass_xchg: a cooperative pass gathers and scatters itself)
    static_assert(sizeof(gc::StoreXchg) % 4 == 0, "exchange record in the word staging");
    const size_t x_off = (io_words + 1) & ~(size_t)1, io_alloc = x_off + sizeof(gc::StoreXchg) / 4;
    const uint32_t turn = s->ser_turn++ & 1u;
    SerArgs a{};
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        int rcs = s->store.flush(ctx);
        hipError_t e = hipSuccess;
        if (rcs == GC_OK && b.h_io_cap < io_alloc) {
            if (b.h_io) (void)hipHostFree(b.h_io);
            b.h_io = nullptr;
            b.h_io_cap = 0;
            e = hipHostMalloc((void **)&b.h_io, (io_alloc + io_alloc / 2) * sizeof(uint32_t), hipHostMallocDefault);
            if (e == hipSuccess) b.h_io_cap = io_alloc + io_alloc / 2;
        }
        if (e == hipSuccess) e = grow(&b.d_io, &b.io_cap, io_alloc);
        if (e == hipSuccess) e = grow(&b.d_boff, &b.boff_cap, (size_t)nblocks + 1);
        // the bytes of this step: 13 header bytes + 3 rows per gate at most
        if (e == hipSuccess) e = grow(&b.d_bytes, &b.bytes_cap, (size_t)ngates * 61 + 16);
        if (e == hipSuccess && !s->ser_stream) e = hipStreamCreateWithFlags(&s->ser_stream, hipStreamNonBlocking);
        for (hipEvent_t *ev : {&s->ev_pass, &s->ser_ev[0], &s->ser_ev[1]})
            if (e == hipSuccess && !*ev) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        if (rcs != GC_OK || e != hipSuccess) {
            gc_circ_release_batch(circ, bt);
            if (e != hipSuccess) set_error("gc_stream_garble", e);
            return rcs != GC_OK ? rcs : e == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
        }
        for (uint32_t i = 0; i < nin; i++) b.h_io[i] = in[i];
        for (uint32_t j = 0; j < nout; j++) {
            b.h_io[nin + j] = out[j];
            b.h_io[nin + nout + j] = s->skip_scratch[j];
        }
        a.gw = circ->d_gwires;
        a.ops = circ->d_ops;
        a.row_of_gate = circ->d_row_of_gate;
        a.in = b.d_io;
        a.out = b.d_io + nin;
        a.ngates = ngates;
        a.first_tmp = first_tmp;
        a.first_out = first_out;
        {
            const gc::StoreXchg x = gc_pass_xchg(circ, s->store.d, b.d_io, b.d_io + nin + nout);
            std::memcpy(b.h_io + x_off, &x, sizeof x);
        }
        e = hipMemcpyAsync(b.d_io, b.h_io, io_alloc * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        // the pass writes bt's tables: behind the serialiser that last read them
        for (int i = 0; i < 2 && e == hipSuccess; i++)
            if (s->ser_batch[i] == bt) e = hipStreamWaitEvent(st, s->ser_ev[i], 0);
        if (e != hipSuccess) {
            gc_circ_release_batch(circ, bt);
            set_error("gc_stream_garble", e);
            return GC_E_HIP;
        }
    }
    tr.lap("uploads");
    // (2) input labels through in[] (Get, :131-141), garble, outputs into the store (Set, :143-157) — all on the device
    int rc = gc_pass_dev(circ, false, s->key.data(), s->key.size(), &s->r, s->store.d, b.d_io, b.d_io + nin + nout, nullptr, 0, &bt,
                         (const gc::StoreXchg *)(b.d_io + x_off));
    if (rc != GC_OK) return rc;
    for (uint32_t j = 0; j < nout; j++)
        if (first_out + j >= first_tmp) s->store.on_dev[out[j]] = 1;
    // (3) byte size of this call's serialisation (depends on in[] / out[]: ids above 0xffff take the long form), then the
    //     wire format (:391-446) written by the device at the scanned offsets — on the serialiser's stream, behind the pass
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipStream_t ss = s->ser_stream;
        hipError_t e = hipEventRecord(s->ev_pass, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(ss, s->ev_pass, 0);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_ser_sizes, dim3(nblocks), dim3(kSerThreads), 0, ss, a, b.d_boff);
            hipLaunchKernelGGL(k_ser_scan, dim3(1), dim3(1024), 0, ss, b.d_boff, nblocks);
            e = hipMemcpyAsync(b.need, b.d_boff + nblocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ss);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_ser_write, dim3(nblocks), dim3(kSerThreads), 0, ss, a, b.d_boff, bt->d_T, bt->g.lt, b.d_bytes);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(b.done, ss);
        if (e == hipSuccess) e = hipEventRecord(s->ser_ev[turn], ss);
        if (e != hipSuccess) {
            set_error("gc_stream_garble", e);
            rc = GC_E_HIP;
            (void)hipStreamSynchronize(ss);
            (void)hipStreamSynchronize(st);
        }
    }
    if (rc == GC_OK) {
        s->ser_batch[turn ^ 1u] = s->ser_batch[turn ^ 1u] == bt ? nullptr : s->ser_batch[turn ^ 1u];
        s->ser_batch[turn] = bt;
        if (s->held) gc_circ_release_batch(s->held_circ, s->held);
        s->held = bt;
        s->held_circ = circ;
    } else {
        gc_circ_release_batch(circ, bt);
    }
    if (rc == GC_OK) {
        b.kind = Slot::kBig;
        b.launched = true;
        s->queue.push_back(StepRef{idx, 0});
        s->n_big_steps++;
    }
    tr.lap("enqueue pass + serialiser");
    return rc;
}
}  // namespace

extern "C" {

// The bytes of the oldest circuit in flight: copied into buf (view == nullptr), or handed out in place (*view = a pointer
// into the engine's pinned staging, valid until the next finish / free call on the stream: the slot it belongs to is only
// given back then)
static int stream_finish(gc_stream *s, uint8_t *buf, size_t cap, size_t *written, const uint8_t **view) {
    if (!s || (!buf && !view) || !written || s->queue.empty()) return GC_E_ARG;
    StreamTrace tr;
    if (s->view_slot != 0xffffffffu) {  // the slot whose bytes the last view pointed into
        Slot &v = *s->slots[s->view_slot];
        if (v.deep_id) s->deep.retire(v.lane, v.deep_id);
        v.reset();
        s->view_slot = 0xffffffffu;
    }
    const StepRef ref = s->queue.front();
    Slot &g = *s->slots[ref.slot];
    while (g.kind == Slot::kGroup && !g.launched) {  // the oldest step sits in an open group: launch up to that one
        int rc = launch_oldest(s);
        if (rc != GC_OK && g.error == GC_OK && g.launched) g.error = rc;
        if (s->win.open.empty()) break;
    }
    s->queue.pop_front();
    gc_ctx *ctx = s->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    int rc = g.error;
    if (rc == GC_OK && !g.synced) {
        // The caller is about to wait for the GPU: nothing new can join an open group meanwhile, so the ctx stream gets the
        // groups behind this one now (up to kKeepQueued in flight) and runs them while the caller digests these bytes.
        // (Launching them any earlier — whenever the stream runs dry — was measured: the groups shrink to 3 - 4 steps and the
        // host's launch sequences become the bound: 1.5e8 against 4.7e8 gates/s on the Ed25519 program.)
        while (!s->win.open.empty() && hipEventQuery(g.done) != hipSuccess && s->ctxq.hungry(s->slots)) {
            int rcq = launch_oldest(s);
            if (rcq != GC_OK) break;
        }
        (void)hipGetLastError();
        hipError_t e = hipEventSynchronize(g.done);
        if (e != hipSuccess) {
            set_error("gc_stream_garble_finish", e);
            rc = GC_E_HIP;
        }
        g.synced = true;
        if (rc == GC_OK) rc = gc_ctx_coop_check(ctx);  // (a cooperative pass that lost a workgroup is noted here)
    }
    if (g.kind == Slot::kGroup) {
        if (rc == GC_OK) {
            const JobRec &j = g.jobs[ref.job];
            const size_t sizes_bytes = up256(g.jobs.size() * sizeof(uint32_t));
            const uint32_t need = ((const uint32_t *)g.h_down)[ref.job];
            *written = need;
            if (view) *view = g.h_down + sizes_bytes + j.off_bytes;
            else if (need > cap) rc = GC_E_ARG;
            else if (need) std::memcpy(buf, g.h_down + sizes_bytes + j.off_bytes, need);
        }
        if (++g.handed == g.jobs.size()) {
            if (view && rc == GC_OK) {
                s->view_slot = ref.slot;  // (given back by the next finish: the caller still reads its bytes)
            } else {
                if (g.deep_id) s->deep.retire(g.lane, g.deep_id);  // (its kernel has run: `done` sits behind it)
                g.reset();
            }
        }
        tr.lap("wait + copy out");
        return rc;
    }
    // a big step: its bytes come straight into the caller's buffer (a view: into the stream's pinned staging)
    if (rc == GC_OK) {
        const uint64_t need = *g.need;
        *written = (size_t)need;
        uint8_t *dst = buf;
        if (view) {
            hipError_t e = grow_pin(ctx, &s->view_buf, &s->view_cap, (size_t)need + 16);
            if (e != hipSuccess) {
                set_error("gc_stream_garble_finish_view", e);
                rc = GC_E_NOMEM;
            }
            dst = s->view_buf;
            *view = dst;
        } else if (need > cap) {
            rc = GC_E_ARG;
        }
        if (rc == GC_OK && need) {  // on its own stream: the next circuit's kernels are already queued on the ctx stream
            hipError_t e = hipMemcpyAsync(dst, g.d_bytes, (size_t)need, hipMemcpyDeviceToHost, s->copy_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(s->copy_stream);
            if (e != hipSuccess) {
                set_error("gc_stream_garble_finish", e);
                rc = GC_E_HIP;
            }
        }
    }
    g.reset();
    tr.lap("wait + d2h");
    return rc;
}

int gc_stream_garble_finish(gc_stream *s, uint8_t *buf, size_t cap, size_t *written) try {
    if (!buf) return GC_E_ARG;
    return stream_finish(s, buf, cap, written, nullptr);
} catch (...) {
    return gc::on_exception();
}

int gc_stream_garble_finish_view(gc_stream *s, const uint8_t **bytes, size_t *len) try {
    if (!bytes) return GC_E_ARG;
    *bytes = nullptr;
    return stream_finish(s, nullptr, 0, len, bytes);
} catch (...) {
    return gc::on_exception();
}

int gc_stream_garble(gc_stream *s, const gc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                     uint32_t nin, const uint32_t *out, uint32_t nout, uint8_t *buf, size_t cap, size_t *written) {
    if (!s || !buf || !written || !s->queue.empty()) return GC_E_ARG;
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
    struct Chunk {       // what the parser records on its way through a block
        uint32_t cmp;    // bytes that must equal the reference block
        uint16_t skip;   // then a global id field (2 / 4 bytes), 0: none
        uint16_t nrows;  // then table rows (16 bytes each)
    };
    size_t nbytes = 0;
    uint32_t nrows = 0, nin = 0, nout = 0;
    CircEntry *ent = nullptr;                // owned by the stream's circuit cache (dropped with it on eviction)
    std::vector<uint8_t> bytes;              // the reference block ...
    std::vector<uint8_t> mask;               // ... and which of its bytes a block of this circuit must repeat (0xff / 0)
    std::vector<uint32_t> row_off;           // byte offset of every table row, in stream order
    std::vector<Chunk> chunks;               // (only in the parser's recording skeleton: build() turns it into mask + row_off)
    std::vector<uint32_t> gf_off;            // global id fields in stream order: byte offset | 1 << 31 for 4-byte ids
    std::vector<uint32_t> gf_canon;          // index of the first field that names the same wire (the repeat pattern)
    std::vector<uint32_t> in_gf, out_gf;     // field of input k (its first read) / of the k-th global write
    std::vector<uint8_t> out_live;           // 0: a later gate of the block writes the same wire (streaming.Set: last wins)
    // the block cut into segments of about equal length, so that several threads can compare / copy side by side (SkelPool)
    struct Seg {
        size_t b0, b1;     // bytes [b0, b1)
        uint32_t r0, r1;   // rows [r0, r1): the rows that start inside [b0, b1)
    };
    std::vector<Seg> segs;
    size_t held() const { return bytes.size() + mask.size() + row_off.size() * sizeof(uint32_t); }
    // mask, row offsets and segments from the parser's chunk list (bytes already holds the block)
    void build(const std::vector<Chunk> &ch) {
        mask.assign(bytes.size(), 0);
        row_off.clear();
        segs.clear();
        const uint32_t n = (uint32_t)ch.size(), per = n / 8 >= 2048 ? n / 8 : n ? n : 1;
        size_t off = 0;
        for (uint32_t c = 0; c < n; c++) {
            if (c % per == 0 && (segs.empty() || n - c >= per / 2)) {
                if (!segs.empty()) segs.back().b1 = off, segs.back().r1 = (uint32_t)row_off.size();
                segs.push_back(Seg{off, bytes.size(), (uint32_t)row_off.size(), 0});
            }
            std::memset(mask.data() + off, 0xff, ch[c].cmp);
            off += (size_t)ch[c].cmp + ch[c].skip;
            for (uint32_t r = 0; r < ch[c].nrows; r++, off += 16) row_off.push_back((uint32_t)off);
        }
        if (segs.empty()) segs.push_back(Seg{0, bytes.size(), 0, 0});
        segs.back().b1 = bytes.size(), segs.back().r1 = (uint32_t)row_off.size();
    }
    // segment sg of `buf` against the reference: equal outside the global ids and the rows?  rows -> slab
    bool match_seg(const Seg &sg, const uint8_t *buf, gc_label *slab) const {
        if (!skel_simd::same(buf + sg.b0, bytes.data() + sg.b0, mask.data() + sg.b0, sg.b1 - sg.b0)) return false;
        if (slab) skel_simd::rows(buf, row_off.data() + sg.r0, sg.r1 - sg.r0, slab + sg.r0);
        return true;
    }
    // the table rows of a block that matched this skeleton with slab == nullptr, into dst (host order)
    void copy_rows(const uint8_t *buf, gc_label *dst) const { skel_simd::rows(buf, row_off.data(), row_off.size(), dst); }
};

// Helper threads for the skeleton match of a big block (a 131 072-gate block is ~33 000 compare runs and 43 000 rows: 0.42 ms
// on one core, more than the GPU needs for the block).  Segments are handed out through a counter; the calling thread
// takes its share.  GC_STREAM_THREADS = helpers (default 3, 0: none).
struct SkelPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    // the job on offer (written under mu; a worker copies it under mu when it wakes up)
    struct Job {
        const EvalSkel *sk = nullptr;
        const uint8_t *buf = nullptr;
        gc_label *slab = nullptr;
        uint32_t id = 0;
    } job;
    bool stop = false;
    // next segment to hand out, tagged with the job it belongs to (id << 32 | index): a worker that was descheduled between
    // its last segment of job k and its next look at the counter must not take — or skip — a segment of job k + 1
    std::atomic<uint64_t> next{0};
    std::atomic<uint32_t> done{0};
    std::atomic<bool> same{true};
    int helpers = -1;

    void work(const Job j) {
        const uint32_t n = (uint32_t)j.sk->segs.size();
        for (;;) {
            uint64_t cur = next.load(std::memory_order_acquire);
            if ((uint32_t)(cur >> 32) != j.id || (uint32_t)cur >= n) return;
            if (!next.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) continue;
            const uint32_t i = (uint32_t)cur;
            if (same.load(std::memory_order_relaxed) && !j.sk->match_seg(j.sk->segs[i], j.buf, j.slab))
                same.store(false, std::memory_order_relaxed);
            // (the job cannot end before this increment: the caller waits for done == n)
            if (done.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
                std::lock_guard<std::mutex> lk(mu);
                cv_done.notify_all();
            }
        }
    }
    void loop() {
        uint32_t seen = 0;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || job.id != seen; });
                if (stop) return;
                j = job;
                seen = j.id;
            }
            work(j);
        }
    }
    bool match(const EvalSkel &s, const uint8_t *b, gc_label *rows) {
        if (helpers < 0) {
            const char *v = std::getenv("GC_STREAM_THREADS");
            helpers = v ? std::atoi(v) : 3;
            helpers = helpers < 0 ? 0 : helpers > 15 ? 15 : helpers;
        }
        if (s.segs.size() < 2 || helpers == 0) {
            for (const EvalSkel::Seg &sg : s.segs)
                if (!s.match_seg(sg, b, rows)) return false;
            return true;
        }
        if (th.empty())
            for (int i = 0; i < helpers; i++) th.emplace_back([this] { loop(); });
        Job j;
        {
            std::lock_guard<std::mutex> lk(mu);
            job.sk = &s, job.buf = b, job.slab = rows;
            job.id++;
            j = job;
            done.store(0, std::memory_order_relaxed);
            same.store(true, std::memory_order_relaxed);
            next.store((uint64_t)j.id << 32, std::memory_order_release);
        }
        cv_work.notify_all();
        work(j);
        const uint32_t n = (uint32_t)s.segs.size();
        if (done.load(std::memory_order_acquire) != n) {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return done.load(std::memory_order_acquire) == n; });
        }
        return same.load();
    }
    ~SkelPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : th) t.join();
    }
};

struct gc_stream_eval {
    gc_ctx *ctx = nullptr;
    std::vector<uint8_t> key;
    int rounds = 0;
    uint32_t *d_rk = nullptr;  // expanded key on the device (step groups)
    DevStore store;  // StreamEval.wires (global store), device-resident
    CircCache cache;
    size_t cache_gates = 0, cache_budget = kCacheGatesDefault;
    uint64_t tick = 0;
    // step groups (see the head of this file): small blocks that share no global wire are evaluated by ONE launch sequence
    std::vector<std::unique_ptr<Slot>> slots;
    GroupWindow win;
    CtxQueue ctxq;
    DeepLanes deep;
    std::vector<gc_label> rows_scratch;  // table rows of a small block while it is parsed
    uint64_t n_groups = 0, n_group_blocks = 0;
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
    // by (ngates, ntmp); a few byte layouts per key, most recently matched first (behind pointers: the order changes often)
    std::unordered_map<uint64_t, std::vector<std::unique_ptr<EvalSkel>>> skels;
    std::vector<uint32_t> gf_ids, wr_ids;   // scratch: the block's global ids by field / the wires it writes
    EvalSkel rec;                           // skeleton of the block being parsed (kept when the parse succeeds)
    bool use_skels = true;                  // GC_STREAM_NO_SKELETON (read at creation): every block is parsed
    SkelPool pool;                          // helper threads of the skeleton match (started by the first big block)
    uint64_t n_parsed = 0, n_matched = 0;
    size_t skel_bytes = 0;                  // reference bytes held by skels (capped: the blocks are the peer's data)
    // table rows of the block being parsed, in pinned memory (true asynchronous H2D); two buffers: the copy of block k
    // may still be in flight while block k + 1 is parsed
    // (kEvalRing buffers: a block's set is re-used once the pass of the block kEvalRing calls ago has run)
    static constexpr uint32_t kEvalRing = 4;
    gc_label *slab_pin[kEvalRing] = {};
    size_t slab_cap[kEvalRing] = {};
    hipEvent_t slab_ev[kEvalRing] = {};
    // ... and its wire maps (inputs, then global outputs): pinned staging + a device copy per ring entry — from pageable
    // memory the upload is synchronous with the stream (the host would wait for the previous block's kernels at every
    // block: the evaluator ran at host time PLUS GPU time per big block), and one device copy would be overwritten under
    // the pass still reading it
    uint32_t *io_pin[kEvalRing] = {};
    uint32_t *io_dev[kEvalRing] = {};
    size_t io_pin_cap[kEvalRing] = {}, io_dev_cap[kEvalRing] = {};
    uint32_t slab_turn = 0;
    // The uploads of a big block (rows, wire maps) run on a stream of their own, under the kernels of the block before:
    // up_ev[i] = this entry's uploads done (the ctx stream waits for it); ring_batch[i] = the pooled batch whose table
    // buffer they went into (an upload into it waits for slab_ev[i]: the pass that last read it); held = the batch of the
    // last block, kept out of the pool until the next block has taken its own (two passes in flight, two table buffers)
    hipStream_t up_stream = nullptr;
    hipEvent_t up_ev[kEvalRing] = {};
    gc_batch *ring_batch[kEvalRing] = {};
    gc_circ *held_circ = nullptr;
    gc_batch *held = nullptr;
};

namespace {

int eval_launch_oldest(gc_stream_eval *e) {
    if (e->win.open.empty()) return GC_OK;
    const uint32_t seq = e->win.first_seq, slot = e->win.pop();
    Slot &g = *e->slots[slot];
    e->n_groups++;
    e->n_group_blocks += g.jobs.size();
    const int rc = launch_group(e->ctx, g, true, e->store, e->d_rk, nullptr, e->rounds, nullptr, e->deep);
    e->win.note(seq, slot, g.launch_no);
    if (rc == GC_OK) e->ctxq.pushed(slot, g.launch_no);
    return rc;
}
int eval_close_group(gc_stream_eval *e) {
    int rc = GC_OK;
    while (!e->win.open.empty()) {
        const int r = eval_launch_oldest(e);
        if (rc == GC_OK) rc = r;
    }
    return rc;
}

// a slot for a new group of blocks: nothing comes back from an evaluator group, so a launched group's slot is free as
// soon as its kernels have run; at most eight groups of the ctx stream in flight, then the oldest is waited for (deep
// blocks on their lanes are bounded by kDeepInFlight and not counted: waiting for a 2 ms multiplier would idle the ctx stream)
Slot *eval_slot(gc_stream_eval *e, uint32_t *index, bool big = false) {
    auto done_with = [&](Slot &sl) {
        if (sl.deep_id) {
            if (sl.error != GC_OK) (void)hipStreamSynchronize(e->deep.lanes[(size_t)sl.lane]);
            e->deep.retire(sl.lane, sl.deep_id);
        }
        sl.reset();
    };
    uint32_t on_ctx = 0;
    bool any_free = false;
    for (auto &sl : e->slots) any_free = any_free || (sl->kind == Slot::kFree && (sl->arena_cap >= ((size_t)2 << 20)) == big);
    if (!any_free) {  // (an event query is a microsecond or two: only when a slot is wanted)
        for (uint32_t i = 0; i < e->slots.size(); i++) {
            Slot &sl = *e->slots[i];
            if (sl.kind == Slot::kGroup && sl.launched && (sl.error != GC_OK || hipEventQuery(sl.done) == hipSuccess)) done_with(sl);
            any_free = any_free || sl.kind == Slot::kFree;
        }
        (void)hipGetLastError();  // hipErrorNotReady of the queries
    }
    for (auto &sl : e->slots)
        if (sl->kind == Slot::kGroup && sl->launched && !sl->deep_id) on_ctx++;
    if (on_ctx >= 8 && !any_free) {
        // the OLDEST launched group (the first one in slot order may be the newest: waiting for that one drains
        // everything queued, and the GPU then idles until the next group is ready — 136 us between the groups of the mixed
        // program)
        Slot *oldest = nullptr;
        for (auto &sl : e->slots)
            if (sl->kind == Slot::kGroup && sl->launched && !sl->deep_id && (!oldest || sl->launch_no < oldest->launch_no)) oldest = sl.get();
        if (oldest) {
            (void)hipEventSynchronize(oldest->done);
            done_with(*oldest);
        }
    }
    return slot_new(e->ctx, e->slots, index, big);
}

}  // namespace

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
        e->rounds = k.rounds;
        e->cache_budget = cache_budget_from_env();
        e->deep.read_env();
        e->use_skels = std::getenv("GC_STREAM_NO_SKELETON") == nullptr;
        hipError_t er = hipSetDevice(ctx->device);
        if (er == hipSuccess) er = hipMalloc((void **)&e->d_rk, sizeof k.w);
        if (er == hipSuccess) er = hipMemcpy(e->d_rk, k.w, sizeof k.w, hipMemcpyHostToDevice);
        if (er != hipSuccess) {
            set_error("gc_stream_eval_create", er);
            rc = GC_E_HIP;
            gc_stream_eval_free(e);
            e = nullptr;
        }
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
    e->deep.release();
    if (e->held) gc_circ_release_batch(e->held_circ, e->held);
    for (auto &kv : e->cache) gc_circ_free(kv.second.circ);
    for (auto &sl : e->slots) sl->release();
    if (e->d_rk) (void)hipFree(e->d_rk);
    if (e->d_io) (void)hipFree(e->d_io);
    if (e->up_stream) {
        (void)hipStreamSynchronize(e->up_stream);
        (void)hipStreamDestroy(e->up_stream);
    }
    for (int i = 0; i < (int)gc_stream_eval::kEvalRing; i++) {
        if (e->up_ev[i]) (void)hipEventDestroy(e->up_ev[i]);
        gc::ctx_buf_put(e->ctx, true, e->slab_pin[i], e->slab_cap[i] * sizeof(gc_label));
        if (e->slab_ev[i]) (void)hipEventDestroy(e->slab_ev[i]);
        gc::ctx_buf_put(e->ctx, true, e->io_pin[i], e->io_pin_cap[i] * sizeof(uint32_t));
        if (e->io_dev[i]) (void)hipFree(e->io_dev[i]);
    }
    e->store.release();
    delete e;
}

int gc_stream_eval_set_wire(gc_stream_eval *e, uint32_t w, const gc_label *l) try {
    if (!e || !l) return GC_E_ARG;
    int rc = eval_close_group(e);  // a queued block reads the wire's OLD label (the reference runs in program order)
    if (rc != GC_OK) return rc;
    if (e->deep.n_inflight) e->deep.drain();  // ... also a deep block on its lane
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

int gc_stream_eval_deep_stats(const gc_stream_eval *e, uint64_t *deep_blocks, uint32_t *lanes) {
    if (!e) return GC_E_ARG;
    if (deep_blocks) *deep_blocks = e->deep.n_steps;
    if (lanes) *lanes = e->deep.state > 0 ? (uint32_t)e->deep.lanes.size() : 0;
    return GC_OK;
}

int gc_stream_eval_get_wire(gc_stream_eval *e, uint32_t w, gc_label *l) try {
    if (!e || !l) return GC_E_ARG;
    int rc = eval_close_group(e);  // a queued block may be the one that writes the wire
    if (rc != GC_OK) return rc;
    if (e->deep.n_inflight) e->deep.drain();  // ... or a deep block on its lane
    rc = e->store.get(e->ctx, w, l);
    (void)gc_ctx_coop_check(e->ctx);  // (the stream was waited for: a cooperative pass that lost a workgroup is noted here)
    return rc;
} catch (...) {
    return gc::on_exception();
}

}  // extern "C"

namespace {

uint32_t stream_max_wires() {
    static const uint32_t v = [] {
        const char *e = std::getenv("GC_STREAM_MAX_WIRES");
        const unsigned long long n = e ? std::strtoull(e, nullptr, 0) : 0;
        return n ? (uint32_t)std::min<unsigned long long>(n, 0xffffffffull) : (1u << 28);
    }();
    return v;
}

// One OpCircuit block (gc_stream_eval_circuit; block after block in gc_stream_eval_blocks)
int eval_block(gc_stream_eval *e, uint32_t ngates, uint32_t ntmp, uint32_t nwires, const uint8_t *buf, size_t len,
               size_t *consumed) {
    *consumed = 0;
    // The block comes from the peer: bound it before anything is sized by it.  A gate is at least 5 bytes (op + two
    // 16-bit ids), so a header that announces more gates than the bytes can hold is a truncated stream; global wire
    // ids must stay below the numWires of the block's own header (the reference indexes its store with them,
    // stream_evaluator.go:29-96: an id beyond it panics there) and tmp ids below numTmpWires.
    // The header's two sizes are the peer's as well, and both size arrays (the reference's InitCircuit allocates numWires and
    // numTmpWires labels on the spot, stream_evaluator.go:52-59): a block of n gates names at most 3 n wires, so a numTmpWires
    // far beyond that is not a compiler's; the global store is bounded by GC_STREAM_MAX_WIRES (default 2^28 wires = 4 GiB of
    // labels on each side of the bus).  (Before the length test: such a block is refused however many of its bytes are here.)
    if ((uint64_t)ntmp > 64ull * ngates + (1u << 20) || nwires > stream_max_wires()) return GC_E_ARG;
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
    // a small block (candidate for a step group) parses its rows into plain host scratch: they are copied into the
    // group's pinned upload region when the block is queued; a big block uses the double-buffered pinned slab
    const bool small_block = ngates <= kSmallGates;
    const uint32_t sb = e->slab_turn % gc_stream_eval::kEvalRing;
    if (small_block) {
        if (e->rows_scratch.size() < (size_t)ngates * 3 + 1) e->rows_scratch.resize((size_t)ngates * 3 + 1);
        GC_HIP(hipSetDevice(e->ctx->device));
    } else {
        e->slab_turn++;
        GC_HIP(hipSetDevice(e->ctx->device));
        if (!e->slab_ev[sb]) GC_HIP(hipEventCreateWithFlags(&e->slab_ev[sb], hipEventDisableTiming));
        else GC_HIP(hipEventSynchronize(e->slab_ev[sb]));  // the pass of the block kEvalRing calls ago (long done)
        const size_t want = (size_t)ngates * 3 + 1;
        if (e->slab_cap[sb] < want) {  // (from the ctx's lists: a stream's evaluator finds the buffers of the one before it)
            uint8_t *p = (uint8_t *)e->slab_pin[sb];
            size_t cap = e->slab_cap[sb] * sizeof(gc_label);
            const hipError_t eg = grow_pin(e->ctx, &p, &cap, want * sizeof(gc_label));
            e->slab_pin[sb] = (gc_label *)p;
            e->slab_cap[sb] = cap / sizeof(gc_label);
            GC_HIP(eg);
        }
    }
    gc_label *slab = small_block ? e->rows_scratch.data() : e->slab_pin[sb];
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
    CircEntry *ent = nullptr;
    const EvalSkel *rows_from = nullptr;  // a matched small block whose rows are still in buf
    uint32_t nin = 0, nout = 0;
    size_t pos = 0;
    std::vector<uint32_t> &gf_ids = e->gf_ids, &wr_ids = e->wr_ids;
    // ---- a block seen before, up to its rows and global ids?
    {
        // the skeleton's bookkeeping for a block that equals it byte-wise: the ids at their known offsets, in the skeleton's
        // repeat pattern?
        auto adopt = [&](const EvalSkel &sk) {
            if (!canon_of(sk.gf_off, &gf_ids, sk.gf_canon.data(), nullptr)) return false;
            ent = sk.ent;
            nin = sk.nin, nout = sk.nout;
            nrows = sk.nrows;
            pos = sk.nbytes;
            e->io_host.resize((size_t)nin + nout + 1);
            for (uint32_t k = 0; k < nin; k++) e->io_host[k] = gf_ids[sk.in_gf[k]];
            wr_ids.resize(nout);
            for (uint32_t k = 0; k < nout; k++) {
                wr_ids[k] = gf_ids[sk.out_gf[k]];
                e->io_host[nin + k] = sk.out_live[k] ? wr_ids[k] : 0xffffffffu;
            }
            // (a small block's rows are copied ONCE, into the upload region of the group it joins — below)
            if (small_block) rows_from = &sk;
            return true;
        };
        auto it = e->skels.end();
        if (e->use_skels && (it = e->skels.find(((uint64_t)ngates << 32) | ntmp)) != e->skels.end()) {
            for (size_t si = 0; si < it->second.size(); si++) {
                const EvalSkel &sk = *it->second[si];
                if (sk.nbytes > len) continue;
                if (!e->pool.match(sk, buf, small_block ? nullptr : slab) || !adopt(sk)) continue;
                // most recently matched first: the variants of a circuit (which operands have 16-bit ids, which repeat) are
                // tried in that order and the least recently matched one goes when there are too many
                if (si) std::rotate(it->second.begin(), it->second.begin() + (long)si, it->second.begin() + (long)si + 1);
                break;
            }
        }
    }
    if (ent) {
        e->n_matched++;
        tr.lap("eval: skeleton match");
    }
    if (!ent) {
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
    ent = cache_find_keys(e->cache, h, gates, cw, nin, nout);
    if (!ent) {
        if (e->cache_gates + ngates + 1 > e->cache_budget && !e->cache.empty()) {
            // The cache is bounded (the blocks are the peer's data: a stream of ever new circuits must not grow host and
            // device memory without limit; the reference evaluator holds one block).  Least recently used circuits go,
            // with the byte skeletons that point at them; nothing may refer to them any more: launch what is queued and
            // drain the stream first (rare: once per budget's worth of NEW circuits).
            int rcq = eval_close_group(e);
            if (rcq != GC_OK) return rcq;
            GC_HIP(hipStreamSynchronize(e->ctx->stream));
            e->deep.drain();
            for (auto &sl : e->slots)
                if (sl->kind == Slot::kGroup && sl->launched) sl->reset();
            if (e->held) gc_circ_release_batch(e->held_circ, e->held);  // (back into its circuit's pool before that may go)
            e->held = nullptr;
            for (auto &rb : e->ring_batch) rb = nullptr;
            cache_make_room(e->cache, &e->cache_gates, e->cache_budget, (size_t)ngates + 1, [&](gc_circ *gone) {
                for (auto &kv : e->skels) {
                    auto &v = kv.second;
                    for (size_t i = v.size(); i-- > 0;)
                        if (v[i]->ent && v[i]->ent->circ == gone) {
                            e->skel_bytes -= std::min(e->skel_bytes, v[i]->held());
                            v.erase(v.begin() + (long)i);
                        }
                }
            });
        }
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
        gc_circ *nc = gc_circ_load(e->ctx, full.data(), ngates, cw, nin, nout, &st);
        if (!nc) return st;
        ent = cache_put_keys(e->cache, h, nc, gates, cw, nin, nout);
        e->cache_gates += ent->cost;
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
    constexpr size_t kSkelVariants = 32;
    constexpr size_t kSkelCap = (size_t)1 << 30;  // beyond 1 GiB of reference blocks every new circuit is parsed each time
    if (e->use_skels && rec_ok && e->skel_bytes + 2 * pos <= kSkelCap) {
        auto &v = e->skels[((uint64_t)ngates << 32) | ntmp];
        // one circuit serialises differently with the widths of the ids it is bound to (per gate: 16-bit ids if all of the
        // gate's are <= 0xffff) and with operands that repeat: an adder of a mixed program shows up in a dozen forms
        if (v.size() >= kSkelVariants) {
            e->skel_bytes -= std::min(e->skel_bytes, v.back()->held());
            v.pop_back();
        }
        std::unique_ptr<EvalSkel> sk(new EvalSkel);
        sk->nbytes = pos;
        sk->nrows = (uint32_t)nrows, sk->nin = nin, sk->nout = nout;
        sk->ent = ent;
        sk->bytes.assign(buf, buf + pos);
        sk->build(rec.chunks);
        sk->gf_off = rec.gf_off;
        sk->in_gf = rec.in_gf, sk->out_gf = rec.out_gf, sk->out_live = rec.out_live;
        if (canon_of(sk->gf_off, &gf_ids, nullptr, &sk->gf_canon)) {
            e->skel_bytes += sk->held();
            v.insert(v.begin(), std::move(sk));
        }
    }
    }  // parsed
    ent->last_use = ++e->tick;
    gc_ctx *ctx = e->ctx;
    // ---- a small block joins the open group: independent blocks are evaluated side by side in one launch sequence; a deep
    //      block (a long one-workgroup pass, DeepLanes) takes a lane ----------------------------------------------------------
    bool is_deep = entry_is_deep(ent, e->deep.min_steps, true) && e->deep.setup(ctx);
    int follow_lane = -1;  // a short block that depends on a deep block in flight follows it onto its lane (see the garbler)
    if (!is_deep && small_block && e->deep.n_inflight && e->deep.follow && entry_is_small(ent)) {
        e->deep.ensure(e->store.host.size());
        follow_lane = e->deep.lane_to_follow(e->deep.conflicts(e->io_host.data(), nin, wr_ids.data(), nout));
        is_deep = follow_lane >= 0;
    }
    if (is_deep || (small_block && entry_is_small(ent))) {
        if (!e->store.dirty.empty()) {  // host-set labels go up before the block's outputs are marked device-owned (and before
            std::lock_guard<std::mutex> lk(ctx->mu);  // the block is put anywhere: a failure leaves nothing half-queued)
            int rcs = e->store.flush(ctx);
            if (rcs != GC_OK) return rcs;
        }
        e->win.ensure(e->store.host.size());
        if (is_deep || e->deep.n_inflight) e->deep.ensure(e->store.host.size());
        const size_t wbytes = up256((size_t)ent->job.w_tile * 16);
        uint32_t gi = e->win.place(e->io_host.data(), nin, wr_ids.data(), nout);
        uint32_t slot_idx = 0;
        if (is_deep) {
            for (; gi > 0; gi--) {  // the open groups this block depends on go first
                int rcq = eval_launch_oldest(e);
                if (rcq != GC_OK) return rcq;
            }
            e->deep.poll();
            if (e->deep.n_inflight >= kDeepInFlight) {
                size_t l = 0;
                for (size_t k = 1; k < e->deep.inflight.size(); k++)
                    if (e->deep.inflight[k].size() > e->deep.inflight[l].size()) l = k;
                (void)hipEventSynchronize(e->deep.inflight[l].front().ev);
                e->deep.poll();
            }
            Slot *ng = eval_slot(e, &slot_idx, follow_lane < 0);
            if (!ng) return GC_E_NOMEM;
            ng->reset();
            ng->kind = Slot::kGroup;
            ng->deep_id = e->deep.new_id();
            ng->lane = follow_lane >= 0 ? follow_lane : e->deep.pick();
            ng->deps = e->deep.conflicts(e->io_host.data(), nin, wr_ids.data(), nout);
            deep_after(e->win, e->slots, e->win.last_conflict(e->io_host.data(), nin, wr_ids.data(), nout), ng);
        } else {
            auto full = [&](const Slot &g) {
                return g.jobs.size() >= kGroupJobs || g.arena_used + g.up_used + wbytes + nrows * 16 > kGroupBytes;
            };
            while (gi < e->win.open.size() && full(*e->slots[e->win.open[gi]])) gi++;
            if (gi == e->win.open.size()) {
                if (e->win.open.size() >= open_groups_limit((size_t)kOpenGroupsMax * 16)) {
                    int rcq = eval_launch_oldest(e);
                    if (rcq != GC_OK) return rcq;
                    gi--;
                }
                uint32_t idx = 0;
                Slot *ng = eval_slot(e, &idx);
                tr.lap("eval: launch + free slot");
                if (!ng) return GC_E_NOMEM;
                ng->reset();
                ng->kind = Slot::kGroup;
                e->win.open.push_back(idx);
            }
            slot_idx = e->win.open[gi];
        }
        Slot &g = *e->slots[slot_idx];
        // the rows of a block of many gates were parsed into the pinned ring: they go up from there (launch_group)
        const bool ext_rows = is_deep && !small_block;
        const size_t io_bytes = up16(((size_t)nin + nout) * sizeof(uint32_t)), row_bytes = ext_rows ? 0 : up16(nrows * sizeof(gc_label));
        hipError_t er = g.reserve_up(up16(g.up_used) - g.up_used + io_bytes + row_bytes + 16);
        if (er != hipSuccess) {
            set_error("gc_stream_eval_circuit (pinned)", er);
            if (is_deep) g.reset();
            return GC_E_NOMEM;
        }
        JobRec j;
        j.ent = ent;
        j.nin = nin, j.nout = nout;
        g.up_used = up16(g.up_used);
        j.off_io = g.up_used;
        if (nin + nout) std::memcpy(g.h_up + g.up_used, e->io_host.data(), ((size_t)nin + nout) * sizeof(uint32_t));
        g.up_used += io_bytes;
        j.off_rows = g.up_used;
        if (!ext_rows && nrows) {
            if (rows_from) rows_from->copy_rows(buf, (gc_label *)(g.h_up + g.up_used));
            else std::memcpy(g.h_up + g.up_used, slab, nrows * sizeof(gc_label));
        }
        g.up_used += row_bytes;
        j.off_w = g.arena_used;
        g.arena_used += wbytes;
        if (ext_rows) {
            // the rows go up NOW, on the upload stream, from the pinned ring entry into the slot's arena: the entry is free
            // again as soon as that copy has run (on the lane it would wait for the deep blocks queued there, and the parser
            // for the entry)
            j.rows_in_arena = true;
            j.off_t = g.arena_used;
            g.arena_used += up256(nrows * sizeof(gc_label));
            std::lock_guard<std::mutex> lk(ctx->mu);
            er = grow_dev(ctx, &g.d_arena, &g.arena_cap, g.arena_used);
            if (er == hipSuccess && !e->up_stream) er = hipStreamCreateWithFlags(&e->up_stream, hipStreamNonBlocking);
            if (er == hipSuccess && !e->up_ev[sb]) er = hipEventCreateWithFlags(&e->up_ev[sb], hipEventDisableTiming);
            if (er == hipSuccess && nrows)
                er = hipMemcpyAsync(g.d_arena + j.off_t, slab, nrows * sizeof(gc_label), hipMemcpyHostToDevice, e->up_stream);
            if (er == hipSuccess) er = hipEventRecord(e->up_ev[sb], e->up_stream);
            (void)hipEventRecord(e->slab_ev[sb], e->up_stream);
            e->ring_batch[sb] = nullptr;
            if (er != hipSuccess) {
                set_error("gc_stream_eval_circuit (rows)", er);
                (void)hipStreamSynchronize(e->up_stream);
                g.reset();
                return er == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
            }
            g.rows_ev = e->up_ev[sb];
        }
        g.lds = std::max(g.lds, ent->lds);
        g.has_or = g.has_or || ent->has_or;
        g.jobs.push_back(j);
        if (!is_deep) {
            e->win.mark(gi, e->io_host.data(), nin, wr_ids.data(), nout);
            if (e->deep.n_inflight) g.deps.merge(e->deep.conflicts(e->io_host.data(), nin, wr_ids.data(), nout));
        }
        if (is_deep) {  // launched at once, on its lane
            int rcl = launch_group(ctx, g, true, e->store, e->d_rk, nullptr, e->rounds, nullptr, e->deep);
            hipStream_t lane = e->deep.lanes[(size_t)g.lane];
            if (rcl != GC_OK) {
                (void)hipStreamSynchronize(lane);
                e->deep.retire(g.lane, g.deep_id);
                g.reset();
                return rcl;
            }
            e->deep.mark(g.deep_id, e->io_host.data(), nin, wr_ids.data(), nout);
            e->n_groups++;
            e->n_group_blocks++;
        }
        for (uint32_t k = 0; k < nout; k++) e->store.on_dev[wr_ids[k]] = 1;
        tr.lap(is_deep ? "eval: launched on a lane" : "eval: queued in group");
        *consumed = pos;
        return GC_OK;
    }
    // ---- a big block: its own launch sequence, behind everything queued -------------------------------------------
    {
        int rcq = eval_close_group(e);
        if (rcq != GC_OK) return rcq;
        if (e->deep.n_inflight) {  // a pass on the ctx stream: behind every deep block in flight (DeepLanes)
            e->deep.poll();
            std::lock_guard<std::mutex> lk(ctx->mu);
            GC_HIP(e->deep.wait_all(ctx->stream));
        }
        {  // ... and later deep blocks must see what it reads and writes
            e->win.ensure(e->store.host.size());
            e->win.mark_pass(e->io_host.data(), nin, wr_ids.data(), nout);
        }
    }
    gc_circ *circ = ent->circ;
    uint32_t *d_io = nullptr;
    const gc::StoreXchg *d_xchg = nullptr;
    gc_batch *b = nullptr;
    if (rows_from) {  // (a matched small block without a one-workgroup plan: rare — its rows into the scratch after all)
        rows_from->copy_rows(buf, slab);
        rows_from = nullptr;
    }
    const gc_label *slab_arg = slab;
    if (!small_block) {
        // the batch (its table buffer) first: not the one of the block before (held), so its pass may still be running
        // while this block's rows are copied in
        int rcb = gc_pass_batch(circ, &b);
        if (rcb != GC_OK) return rcb;
        slab_arg = nullptr;
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipError_t er = hipSetDevice(ctx->device);
        int rcs = er == hipSuccess ? e->store.flush(ctx) : GC_E_HIP;
        const size_t nio = e->io_host.size();
        const size_t x_off = (nio + 1) & ~(size_t)1, nio_alloc = x_off + sizeof(gc::StoreXchg) / 4;
        if (rcs != GC_OK) {
        } else if (small_block) {  // (without an LDS plan: rare; the pass is waited for below)
            er = grow(&e->d_io, &e->io_cap, nio);
            if (er == hipSuccess) er = hipMemcpyAsync(e->d_io, e->io_host.data(), nio * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
            d_io = e->d_io;
        } else {
            // this ring entry's pinned staging and device copy (free: slab_ev[sb] was waited for above); rows and wire maps
            // go up on the upload stream, behind the last pass that read this table buffer, and the ctx stream waits for them
            if (e->io_pin_cap[sb] < nio_alloc) {
                uint8_t *p = (uint8_t *)e->io_pin[sb];
                size_t cap = e->io_pin_cap[sb] * sizeof(uint32_t);
                er = grow_pin(ctx, &p, &cap, nio_alloc * sizeof(uint32_t));
                e->io_pin[sb] = (uint32_t *)p;
                e->io_pin_cap[sb] = cap / sizeof(uint32_t);
            }
            if (er == hipSuccess) er = grow(&e->io_dev[sb], &e->io_dev_cap[sb], nio_alloc);
            if (er == hipSuccess && !e->up_stream) er = hipStreamCreateWithFlags(&e->up_stream, hipStreamNonBlocking);
            if (er == hipSuccess && !e->up_ev[sb]) er = hipEventCreateWithFlags(&e->up_ev[sb], hipEventDisableTiming);
            for (uint32_t i = 0; i < gc_stream_eval::kEvalRing && er == hipSuccess; i++)
                if (e->ring_batch[i] == b && i != sb && e->slab_ev[i]) er = hipStreamWaitEvent(e->up_stream, e->slab_ev[i], 0);
            if (er == hipSuccess && nrows)
                er = hipMemcpyAsync(b->d_T, slab, nrows * sizeof(gc_label), hipMemcpyHostToDevice, e->up_stream);
            if (er == hipSuccess) {
                std::memcpy(e->io_pin[sb], e->io_host.data(), nio * sizeof(uint32_t));
                // ... and the pass's label exchange record behind them (a cooperative pass gathers and scatters itself)
                const gc::StoreXchg x = gc_pass_xchg(circ, e->store.d, e->io_dev[sb], e->io_dev[sb] + nin);
                std::memcpy(e->io_pin[sb] + x_off, &x, sizeof x);
                d_xchg = (const gc::StoreXchg *)(e->io_dev[sb] + x_off);
                er = hipMemcpyAsync(e->io_dev[sb], e->io_pin[sb], nio_alloc * sizeof(uint32_t), hipMemcpyHostToDevice, e->up_stream);
            }
            if (er == hipSuccess) er = hipEventRecord(e->up_ev[sb], e->up_stream);
            if (er == hipSuccess) er = hipStreamWaitEvent(ctx->stream, e->up_ev[sb], 0);
            d_io = e->io_dev[sb];
        }
        if (rcs != GC_OK || er != hipSuccess) {
            if (er != hipSuccess) set_error("gc_stream_eval_circuit", er);
            if (b) {
                // a copy into the batch's buffers may be in flight: nothing else may use it before that has run
                (void)hipStreamSynchronize(e->up_stream);
                gc_circ_release_batch(circ, b);
            }
            return rcs != GC_OK ? rcs : er == hipErrorOutOfMemory ? GC_E_NOMEM : GC_E_HIP;
        }
    }
    int rc = gc_pass_dev(circ, true, e->key.data(), e->key.size(), nullptr, e->store.d, d_io, d_io + nin, slab_arg, nrows, &b, d_xchg);

    // the pinned slab (and this entry's buffers) may be overwritten once what was enqueued from it has run: also when the
    // pass failed half-way (the uploads are in flight)
    if (!small_block) {
        (void)hipEventRecord(e->slab_ev[sb], ctx->stream);
        for (auto &rb : e->ring_batch)
            if (rb == b) rb = nullptr;
        e->ring_batch[sb] = b;  // (null when the pass failed: gc_pass_dev has put the batch back)
    } else {
        (void)hipStreamSynchronize(ctx->stream);  // rows in plain host scratch (a small block without an LDS plan: rare)
    }
    if (rc != GC_OK) {
        if (!small_block) (void)hipStreamSynchronize(e->up_stream);
        return rc;
    }
    if (small_block) {
        gc_circ_release_batch(circ, b);
    } else {
        if (e->held) gc_circ_release_batch(e->held_circ, e->held);
        e->held = b;
        e->held_circ = circ;
    }
    for (uint32_t k = 0; k < nout; k++) e->store.on_dev[e->wr_ids[k]] = 1;
    tr.lap("eval: enqueue");
    *consumed = pos;
    return GC_OK;
}

}  // namespace

extern "C" {

int gc_stream_eval_circuit(gc_stream_eval *e, uint32_t ngates, uint32_t ntmp, uint32_t nwires, const uint8_t *buf,
                           size_t len, size_t *consumed) try {
    if (!e || (!buf && len) || !consumed) return GC_E_ARG;
    return eval_block(e, ngates, ntmp, nwires, buf, len, consumed);
} catch (...) {
    return gc::on_exception();
}

int gc_stream_eval_blocks(gc_stream_eval *e, const uint8_t *buf, size_t len, size_t *consumed, uint32_t *nblocks,
                          int *more) try {
    if (!e || (!buf && len) || !consumed) return GC_E_ARG;
    *consumed = 0;
    if (nblocks) *nblocks = 0;
    if (more) *more = 0;
    auto be32 = [](const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; };
    size_t pos = 0;
    uint32_t n = 0;
    int rc = GC_OK;
    bool cut = false;
    while (len - pos >= 20 && be32(buf + pos) == 1 /* OpCircuit, stream_evaluator.go:22-26 */) {
        const uint32_t ngates = be32(buf + pos + 8), ntmp = be32(buf + pos + 12), nwires = be32(buf + pos + 16);
        const uint8_t *body = buf + pos + 20;
        const size_t avail = len - pos - 20;
        // The last block of a buffer is, as a rule, cut off.  Finding that out by parsing it gate by gate until the bytes run
        // out costs more than evaluating a whole block: when every known layout of this (gates, tmp wires) key is longer
        // than what is here and what is here equals the head of one of them, the block is a known one that is not complete.
        // (A NEW layout that is complete in fewer bytes and shares that head is taken for incomplete as well: the caller
        // comes back with more bytes — in a well-formed stream something always follows a block — and it is parsed then.)
        if ((uint64_t)ntmp > 64ull * ngates + (1u << 20) || nwires > stream_max_wires()) {
            rc = GC_E_ARG;  // (eval_block's own test, in front of the look at what is here of the block)
            break;
        }
        auto it = e->use_skels ? e->skels.find(((uint64_t)ngates << 32) | ntmp) : e->skels.end();
        if (it != e->skels.end()) {
            bool fits = false, head = false;
            for (const auto &sk : it->second) fits = fits || sk->nbytes <= avail;
            if (!fits)
                for (const auto &sk : it->second) head = head || skel_simd::same(body, sk->bytes.data(), sk->mask.data(), avail);
            if (head) {
                cut = true;
                break;
            }
        }
        size_t used = 0;
        rc = eval_block(e, ngates, ntmp, nwires, body, avail, &used);
        if (rc != GC_OK) break;
        pos += 20 + used;
        n++;
    }
    // a block that ends beyond the buffer is not an error of this call: the caller brings more bytes, or knows that none come
    if (rc == GC_E_ROWS) {
        cut = true;
        rc = GC_OK;
    }
    if (rc == GC_OK && pos < len && (len - pos < 4 || (len - pos < 20 && be32(buf + pos) == 1))) cut = true;  // a header cut in two
    if (more && cut) *more = 1;
    *consumed = pos;
    if (nblocks) *nblocks = n;
    return rc;
} catch (...) {
    return gc::on_exception();
}

}  // extern "C"
