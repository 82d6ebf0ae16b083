// stream_internal.h — what the translation units of the streaming engine share (stream_garble.cpp, stream_eval.cpp,
// stream_group.cpp, stream_serialise.cpp, stream_lanes.cpp): the circuit cache, the device-resident wire store, launch
// slots, the window of open groups, the deep lanes and the argument records of the device-side serialiser.  Internal:
// nothing here crosses the C ABI.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <new>
#include <thread>
#include <unordered_map>

#include "engine.h"

namespace gcs {
using namespace gc;

// developer aid: where the host's share of a stream goes (cycle counters per stage, printed when the stream is freed under
// GC_TRACE; a reading costs ~20 cycles)
struct StageProf {
    enum { kGuess, kWait, kAdopt, kPlace, kQueue, kMark, kLaunch, kFinish, kOther, kN };
    uint64_t cyc[kN] = {}, last = 0;
    static uint64_t now() { return __builtin_ia32_rdtsc(); }
    void start() { last = now(); }
    void lap(int stage) {
        const uint64_t t = now();
        cyc[stage] += t - last;
        last = t;
    }
    void print(const char *who, uint64_t steps) const {
        if (!std::getenv("GC_TRACE") || !steps) return;
        static const char *names[kN] = {"guess", "wait", "adopt", "place", "queue", "mark", "launch", "finish", "other"};
        uint64_t tot = 0;
        for (uint64_t c : cyc) tot += c;
        std::fprintf(stderr, "[gc trace] %s: host cycles per step (%llu steps):", who, (unsigned long long)steps);
        for (int i = 0; i < kN; i++) std::fprintf(stderr, " %s %.0f", names[i], (double)cyc[i] / (double)steps);
        std::fprintf(stderr, " | total %.0f\n", (double)tot / (double)steps);
    }
};

// developer aid: GC_TRACE=2 prints the wall-clock laps of every streaming step to stderr (GC_TRACE=1: only the summaries — a
// line per step distorts what the stage cycles above measure)
struct StreamTrace {
    static bool enabled() {  // (asked once: two getenv walks per streamed step were measurable on 512-gate steps)
        static const bool v = std::getenv("GC_TRACE") != nullptr && std::atoi(std::getenv("GC_TRACE")) >= 2;
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

// developer aid: GC_TRACE=3 — a timeline of the garbler's step groups: per group the host's clock when its launch sequence was
// issued and when the caller waited for its bytes, and the GPU's clock (timing events) at the head of the sequence, behind
// the kernel and behind the copy of the bytes; printed when the stream is freed (scripts/group_timeline.py reads it)
struct GroupTimeline {
    static bool enabled() {
        static const bool v = std::getenv("GC_TRACE") != nullptr && std::atoi(std::getenv("GC_TRACE")) == 3;
        return v;
    }
    struct G {
        uint64_t launch_no = 0;
        uint32_t steps = 0;
        double h_l0 = 0, h_l1 = 0, h_w0 = -1, h_w1 = -1;
        hipEvent_t k0 = nullptr, k1 = nullptr, d = nullptr;
    };
    std::vector<G> groups;
    hipEvent_t ref = nullptr;
    std::chrono::steady_clock::time_point t0;
    double now() const { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
    G *find(uint64_t launch_no) {
        for (size_t i = groups.size(); i-- > 0 && groups.size() - i < 64;)
            if (groups[i].launch_no == launch_no) return &groups[i];
        return nullptr;
    }
    void print_and_clear();
};
GroupTimeline &group_timeline();

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
    // chain fusion (stream_fuse.cpp): the ctx-wide identity of the circuit's content (0: not registered — never part of a
    // fused chain), whether `gates` holds the evaluator's parsed form, and the gates written in the wire format's own ids
    uint32_t uid = 0;
    bool eval_form = false;
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

// Chain fusion (stream_fuse.cpp): a queued step whose dependencies inside the window all sit in ONE job of the latest group
// it conflicts with is appended to that job instead of opening a later group — the chain (mul -> add -> add -> ... -> carry)
// then runs as ONE planned circuit whose gates are scheduled across the step boundaries.
constexpr uint32_t kFuseTailGates = 1024;   // a step of at most this many gates may be appended to a chain ...
constexpr uint32_t kFuseMembers = 48;       // ... of at most this many steps,
constexpr uint32_t kFuseGates = 40960;      //     gates,
constexpr uint32_t kFuseSlots = 2600;       //     live labels (the sum of the members' own LDS plans: an estimate),
constexpr uint32_t kFuseInputs = 2048;      //     and labels read from the wire store
constexpr uint32_t kFuseDepth = 320;        // ... and dependent hash phases, as far as earlier chains of the same shape tell (fuse_depth_hint;
                                            //     GC_STREAM_FUSE_DEPTH overrides it); for a shape nobody has planned yet: twice that as the
                                            //     sum of the steps' own depths (no overlap assumed)
inline uint32_t fuse_depth_cap() {
    static const uint32_t v = [] {
        const char *e = std::getenv("GC_STREAM_FUSE_DEPTH");
        const int n = e && *e ? std::atoi(e) : 0;
        return n > 0 ? (uint32_t)n : kFuseDepth;
    }();
    return v;
}
constexpr uint32_t kGroupSteps = 4096;      // steps per group at most (fused or not)
constexpr uint32_t kFuseMulti = 0xfffffffeu, kFuseNone = 0xffffffffu;
// Dependent units in ONE launch (kernels.h: launch_fused_flat_jobs, d_sync) — an EXPERIMENT, off unless GC_STREAM_DEPS=1 is in
// the environment when the stream is created (EXPERIMENTS.md, round 5: it did not pay).  A step whose conflicts with the
// latest open group it has any with cannot be fused into one unit of that group still JOINS that group, as a unit that waits
// — on the device, by done-flags — for the units it conflicts with, instead of waiting for the whole group in the next
// launch.  A group is then as long as the longest dependency path inside it, not the sum of its longest steps level by
// level.  kUnitDeps: the units a step may name; beyond that (or where the window's records do not tell them apart: a wire
// several units read) it waits for every earlier unit.
constexpr uint32_t kUnitDeps = 8;
constexpr uint32_t kDepsAllEarlier = 0xffffffffu;
inline bool deps_wanted() {
    const char *e = std::getenv("GC_STREAM_DEPS");
    return e && *e && std::strcmp(e, "0") != 0;
}

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
    // chain fusion: the launch unit (Slot::wgs) this step belongs to, the next step of the same unit (-1: the last) and, for
    // every step but a unit's first, where Slot::wiring holds the source of each of its inputs (kFuseNone: read from the wire
    // store; else member << 24 | output index of an earlier step of the unit)
    uint32_t wg = 0, member = 0;  // (member: its position in the unit)
    int32_t next = -1;
    size_t off_wiring = 0;
    uint32_t nrows = 0;           // evaluator: table rows of the block
    // evaluator, a block the DEVICE matched (stream_eval_dev.cpp): its bytes are in device memory — a chunk of the peer's
    // stream — and the launch sequence gathers its rows from there (d_row_off: the skeleton's row offsets, a device array)
    // into the job's table array (arena, off_t)
    const uint8_t *d_block = nullptr;
    const uint32_t *d_row_off = nullptr;
    uint32_t chunk = 0;
};

// one launch unit of a group = one workgroup of its kernel: ONE step, or a chain of dependent steps (chain fusion)
struct WgRec {
    uint32_t head = 0, tail = 0, n = 0;   // steps of the unit (indices into Slot::jobs), linked through JobRec::next
    uint32_t gates = 0, slots = 0, inputs = 0, depth_sum = 0;  // what the caps of the fusion count
    uint64_t shape = 0;                   // running hash of (circuits, wiring) of its steps: what the depth hints are kept by
    bool open = true;                     // may take further steps (a unit whose head cannot be fused is closed from the start)
    uint32_t dep_off = 0, dep_n = 0;      // units of the same group it waits for: Slot::unit_deps[dep_off .. dep_off + dep_n)
                                          // (dep_n == kDepsAllEarlier: every unit before it)
    uint64_t anc[4] = {0, 0, 0, 0};       // ... and everything those wait for in turn: the units that are DONE when this one starts
    static_assert(kGroupJobs <= 256, "one bit per unit of a group");
    bool behind(uint32_t u) const { return (anc[(u >> 6) & 3u] >> (u & 63u)) & 1u; }
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

// A few threads that copy bytes for the stream's thread (the garbler's copies of finished steps into the caller's buffer: 2.3 GB
// for the Ed25519-shaped program, more than one core moves while it also queues 26 000 steps).  Tasks are (dst, src, len) with a
// counter to decrement when done; large ones are cut into pieces so that the threads share them.  wait() returns when every
// task handed over so far has been done.
class CopyPool {
public:
    ~CopyPool() { stop(); }
    void submit(void *dst, const void *src, size_t len, std::atomic<uint32_t> *pending) {
        if (th_.empty()) start();
        constexpr size_t kPiece = (size_t)256 << 10;
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t off = 0; off < len || off == 0; off += kPiece) {
            const size_t n = std::min(kPiece, len - off);
            if (pending) pending->fetch_add(1, std::memory_order_relaxed);
            q_.push_back(Task{(uint8_t *)dst + off, (const uint8_t *)src + off, n, pending});
            issued_++;
            if (len == 0) break;
        }
        // (a wake-up is a system call: only when a thread is really asleep — with 26 000 copies a run, most find them all busy)
        if (idle_ >= 2 && q_.size() >= 2) cv_.notify_all();
        else if (idle_) cv_.notify_one();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return done_ == issued_; });
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
        th_.clear();
        quit_ = false;
    }

private:
    struct Task {
        uint8_t *dst;
        const uint8_t *src;
        size_t len;
        std::atomic<uint32_t> *pending;
    };
    void start() {
        const char *v = std::getenv("GC_STREAM_COPY_THREADS");
        int n = v && *v ? std::atoi(v) : 3;
        n = n < 1 ? 1 : n > 8 ? 8 : n;
        for (int i = 0; i < n; i++) th_.emplace_back([this] { loop(); });
    }
    void loop() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                idle_++;
                cv_.wait(lk, [&] { return quit_ || !q_.empty(); });
                idle_--;
                if (q_.empty()) return;
                t = q_.front();
                q_.pop_front();
            }
            if (t.len) std::memcpy(t.dst, t.src, t.len);
            if (t.pending) t.pending->fetch_sub(1, std::memory_order_release);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (++done_ == issued_) done_cv_.notify_all();
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<Task> q_;
    std::vector<std::thread> th_;
    uint64_t issued_ = 0, done_ = 0;
    uint32_t idle_ = 0;
    bool quit_ = false;
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
    // gc_stream_garble_finish_async: copies out of the slot's pinned bytes still under way on the copier threads; a slot whose
    // steps have all been handed out is only given back (retire) when they are done
    std::atomic<uint32_t> copies{0};
    bool retire = false;
    uint32_t pool_id = 0xffffffffu, pool_done = 0;  // Dataflow (pool): the slot's counter in PoolCtl::done and what it has reached (kept across reset())
    hipEvent_t kdone = nullptr, done = nullptr;  // kernels of the group enqueued-and-done / bytes back in pinned memory
    hipEvent_t kernel_ev = nullptr;              // whichever of the two says "the group's kernel has run" (set at launch)
    std::vector<JobRec> jobs;
    std::vector<WgRec> wgs;          // launch units, in the order their first steps were queued
    std::vector<uint32_t> wiring;    // chain fusion: input sources of the appended steps (JobRec::off_wiring)
    // ... and outputs a LATER step of the same unit writes again: (step, output index) pairs whose store is dropped (in a
    // fused job every output goes back to the wire store at the end, side by side: the last writer must be the only one)
    std::vector<std::pair<uint32_t, uint32_t>> kills;
    std::vector<uint32_t> unit_deps;  // the units that the waiting units of the group wait for (WgRec::dep_off / dep_n)
    bool has_waits = false;
    std::vector<uint32_t> chunk_refs;  // evaluator: the chunks of the peer's stream the slot's jobs gather their rows from (one per job)
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
        retire = false;
        deep_id = 0;
        deps = DeepDeps{};
        after_tail = false;
        after_ev = nullptr;
        rows_ev = nullptr;
        lane = -1;
        jobs.clear();
        wgs.clear();
        wiring.clear();
        kills.clear();
        unit_deps.clear();
        has_waits = false;
        chunk_refs.clear();
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
inline hipError_t grow_buf(gc_ctx *ctx, bool pinned, uint8_t **p, size_t *cap, size_t need) {
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
inline hipError_t grow_dev(gc_ctx *ctx, uint8_t **p, size_t *cap, size_t need) { return grow_buf(ctx, false, p, cap, need); }
inline hipError_t grow_pin(gc_ctx *ctx, uint8_t **p, size_t *cap, size_t need) { return grow_buf(ctx, true, p, cap, need); }

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
// results: the upper bound.
// Round 5: with chain fusion the links of such a chain are ONE launch unit of one group, the window no longer needs a group
// per link — and sixteen open groups hold everything a caller with 1 024 in flight allows (~7 groups of ~137 steps), so groups
// went to the GPU only when the caller asked for bytes, and caller and GPU took turns waiting (host stage cycles: 30 - 45 % of
// the garbler's run in hipEventSynchronize with the GPU 28 % busy).  At most 6 open groups while fusion is on: the Ed25519-shaped
// program, C host, bytes in place 0.94 - 1.02e9 -> 1.36 - 1.38e9 gates/s, copied by the copier threads 0.70 -> 0.89 - 1.01e9, the
// evaluator block by block 3.4 -> 4.0 - 4.4e8 (5: the same; 7, 10, 12: as 16).
constexpr uint32_t kOpenGroupsMin = 4, kOpenGroupsMax = 16, kOpenGroupsMaxFused = 6;
bool fuse_enabled();
inline uint32_t open_groups_env() {
    static const uint32_t v = [] {
        const char *e = std::getenv("GC_STREAM_OPEN_GROUPS");
        const int n = e && *e ? std::atoi(e) : 0;
        return (uint32_t)std::min(std::max(n, 0), 48);
    }();
    return v;
}
inline uint32_t open_groups_limit(size_t in_flight) {
    if (open_groups_env()) return open_groups_env();
    return (uint32_t)std::min<size_t>(std::max<size_t>(in_flight / 16, kOpenGroupsMin), fuse_enabled() ? kOpenGroupsMaxFused : kOpenGroupsMax);
}
// ---- dataflow between launch units ACROSS launches (round 6, GC_STREAM_DATAFLOW=1, the garbler; kernels.h: DfBlock) ------------
// Today a group is a level: it ends with its slowest unit and the ctx stream runs the groups one after the other; the ideal
// under the caller's window is twice what that reaches on the instruction mixes (scripts/stream_ideal_model.py).  Here the wire
// store carries per wire how often launch units have WRITTEN (ver) and READ (rd) it; the host keeps the same counts in launch
// order — which is program order for every two units that share a wire: that is the invariant the windows, lanes and events
// already keep — and every unit is told what to wait for (launch_group builds its DfBlock).  The groups of the window are
// formed exactly as before (chain fusion included) but launched on ROTATING streams with no order between them; what is not a
// group launch (big steps, uploads of host-set labels, read-backs) joins all of them first and bumps the counts itself.
inline uint32_t df_streams_wanted() {  // (GC_STREAM_DF_STREAMS: how many streams the group launches rotate over; default 4)
    const char *e = std::getenv("GC_STREAM_DF_STREAMS");
    const int n = e && *e ? std::atoi(e) : 4;
    return (uint32_t)std::min(std::max(n, 1), 12);
}
// Launch units launched and not known to be done, at most PER XCD: a unit that waits holds its CU, so every unit launched must
// be able to become resident beside the others — and workgroup i of EVERY launch goes to XCD i mod 8 (32 CUs each): a hundred
// launches of one unit would all sit on XCD 0.
constexpr uint32_t kDfUnitsPerXcd = 26;
inline uint32_t pool_workers_wanted() {  // (GC_STREAM_DF_WORKERS: persistent workgroups of GC_STREAM_DATAFLOW=4; default 96 of 256 CUs)
    const char *e = std::getenv("GC_STREAM_DF_WORKERS");
    const int n = e && *e ? std::atoi(e) : 96;
    return (uint32_t)std::min(std::max(n, 1), 224);
}
#define kPoolWorkers pool_workers_wanted()
inline bool dataflow_wanted() {
    const char *e = std::getenv("GC_STREAM_DATAFLOW");
    return e && *e && std::strcmp(e, "0") != 0;
}
struct Slot;
struct Dataflow {
    bool on = false;
    struct VR {
        uint32_t ver, rd;
    };
    std::vector<VR> vr;             // host mirrors of the device counts, advanced at launch (side by side: one cache line per wire)
    uint32_t *d_ver = nullptr, *d_rd = nullptr;
    size_t cap = 0;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> tail;   // per stream: behind its latest launch
    std::vector<uint8_t> used;      // ... if it has had one since the last join
    uint32_t turn = 0;
    hipEvent_t ctx_ev = nullptr;    // behind the latest pass of the ctx stream that touched the store outside a group launch
    bool ctx_ev_set = false;
    struct InFlight {
        Slot *slot;
        uint64_t launch_no;
        uint32_t units;
    };
    // out-of-order issue on top (GC_STREAM_DATAFLOW=3; kernels.h: PoolCtl): the units of every launch are published into a ring
    // by the upload stream, in launch order, and the workgroups of any launch claim the lowest unclaimed one
    bool pool = false;
    gc::PoolCtl *d_pool = nullptr;
    hipStream_t up_stream = nullptr;
    uint32_t published = 0;   // tickets published so far
    uint32_t next_group = 0;  // ids handed to slots (Slot::pool_id: the slot's counter in PoolCtl::done)
    // =4: PERSISTENT workgroups (PoolCtl::persist): one launch of kPoolWorkers on a stream of its own serves every publication
    // until the next join raises `stop`
    bool persist = false, pool_running = false;
    int pool_rounds = 14;
    hipStream_t pool_stream = nullptr;
    hipEvent_t pool_ev = nullptr, pool_go = nullptr;
    uint64_t n_pool_starts = 0;
    std::deque<InFlight> inflight;
    uint32_t xcd_load[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // workgroups in flight per XCD (workgroup i of a launch: XCD i mod 8)
    uint64_t n_launches = 0, n_cap_waits = 0;
    static uint32_t on_xcd(uint32_t units, uint32_t x) { return (units + 7u - x) / 8u; }
    bool fits(uint32_t units) const {
        for (uint32_t x = 0; x < 8; x++)
            if (xcd_load[x] + on_xcd(units, x) > kDfUnitsPerXcd) return false;
        return true;
    }
    void account(uint32_t units, bool add) {
        for (uint32_t x = 0; x < 8; x++) {
            const uint32_t n = on_xcd(units, x);
            xcd_load[x] = add ? xcd_load[x] + n : xcd_load[x] - std::min(xcd_load[x], n);
        }
    }

    hipError_t setup() {
        if (!streams.empty()) return hipSuccess;
        const char *mode = std::getenv("GC_STREAM_DATAFLOW");
        // (persistent workgroups: no launch has workgroups of its own — one stream for the order of its waits is enough, and every
        // stream less is one less that may share a hardware queue with the kernel that never ends)
        for (uint32_t k = 0, n = mode && mode[0] == '4' ? 1u : df_streams_wanted(); k < n; k++) {
            hipStream_t st = nullptr;
            hipEvent_t ev = nullptr;
            hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) return e;
            streams.push_back(st), tail.push_back(ev), used.push_back(0);
        }
        hipError_t e = hipEventCreateWithFlags(&ctx_ev, hipEventDisableTiming);
        const char *m = std::getenv("GC_STREAM_DATAFLOW");
        if (e == hipSuccess && m && (m[0] == '3' || m[0] == '4')) {
            e = hipStreamCreateWithFlags(&up_stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipMalloc((void **)&d_pool, sizeof(gc::PoolCtl));
            if (e == hipSuccess) e = hipMemset(d_pool, 0, sizeof(gc::PoolCtl));
            // (persistent workgroups never leave their hardware queue: any stream of the process that SHARES that queue would wait for
            // them for ever — publications included.  The runtime multiplexes streams onto GPU_MAX_HW_QUEUES queues (4 by default);
            // the host must have asked for at least 16, else the form with workgroups per launch, =3, is used)
            const char *q = std::getenv("GPU_MAX_HW_QUEUES");
            if (e == hipSuccess && m[0] == '4' && q && std::atoi(q) >= 16) {
                const uint32_t one = 1;
                e = hipMemcpy(&d_pool->persist, &one, sizeof one, hipMemcpyHostToDevice);
                if (e == hipSuccess) e = hipStreamCreateWithFlags(&pool_stream, hipStreamNonBlocking);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&pool_ev, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&pool_go, hipEventDisableTiming);
                persist = e == hipSuccess;
            }
            if (e == hipSuccess) e = hipDeviceSynchronize();  // (the fill runs on the null stream: the streams here do not wait for it)
            pool = e == hipSuccess;
        }
        return e;
    }
    // the counts cover wires [0, n): a grown array starts at zero for the new wires (nothing may be in flight: the caller's
    // store has just moved with a device-wide wait, or this is the first use)
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        const size_t ncap = std::max(n, cap * 2);
        uint32_t *nv = nullptr, *nr = nullptr;
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMalloc((void **)&nv, ncap * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&nr, ncap * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(nv, 0, ncap * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(nr, 0, ncap * sizeof(uint32_t));
        if (e == hipSuccess && cap) e = hipMemcpy(nv, d_ver, cap * sizeof(uint32_t), hipMemcpyDeviceToDevice);
        if (e == hipSuccess && cap) e = hipMemcpy(nr, d_rd, cap * sizeof(uint32_t), hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipDeviceSynchronize();  // (fills and copies of the null stream: the non-blocking streams do not wait for them)
        if (e != hipSuccess) {
            if (nv) (void)hipFree(nv);
            if (nr) (void)hipFree(nr);
            return e;
        }
        if (d_ver) (void)hipFree(d_ver);
        if (d_rd) (void)hipFree(d_rd);
        d_ver = nv, d_rd = nr, cap = ncap;
        vr.resize(ncap, VR{0, 0});
        return hipSuccess;
    }
    hipStream_t next(uint32_t *k) {
        *k = turn++ % (uint32_t)streams.size();
        return streams[*k];
    }
    // `st` (the ctx stream, as a rule) waits for every group launched so far
    // persistent workgroups: make sure they run (behind a `stop` of zero), before a publication
    hipError_t pool_start(size_t lds_bytes) {
        if (!persist || pool_running) return hipSuccess;
        hipError_t e = hipMemsetAsync(&d_pool->stop, 0, sizeof(uint32_t), up_stream);
        if (e == hipSuccess) e = hipEventRecord(pool_go, up_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(pool_stream, pool_go, 0);
        if (e == hipSuccess) e = gc::launch_fused_flat_pool(pool_rounds, d_pool, kPoolWorkers, lds_bytes, pool_stream);
        if (e == hipSuccess) e = hipEventRecord(pool_ev, pool_stream);
        pool_running = e == hipSuccess;
        n_pool_starts++;
        return e;
    }
    // ... and have left (behind everything published so far) before `st` goes on
    hipError_t pool_stop(hipStream_t st) {
        if (!persist || !pool_running) return hipSuccess;
        gc::launch_pool_stop(d_pool, up_stream);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && st) e = hipStreamWaitEvent(st, pool_ev, 0);
        pool_running = false;
        return e;
    }
    hipError_t join(hipStream_t st) {
        hipError_t e = hipSuccess;
        if (pool) {  // every unit published so far has counted itself done (whoever ran it)
            if (published) gc::launch_pool_wait(&d_pool->done_total, published, nullptr, st);
            e = hipGetLastError();
            if (e == hipSuccess) e = pool_stop(st);  // (a big step's cooperative launch wants an XCD's CUs to itself)
            return e;
        }
        for (size_t k = 0; k < streams.size() && e == hipSuccess; k++)
            if (used[k]) e = hipStreamWaitEvent(st, tail[k], 0);
        return e;
    }
    // ... and every later group launch for what `st` holds now
    hipError_t fence_from(hipStream_t st) {
        hipError_t e = hipEventRecord(ctx_ev, st);
        ctx_ev_set = e == hipSuccess;
        return e;
    }
    void drain() {
        if (pool && up_stream) {
            (void)hipStreamSynchronize(up_stream);
            if (published) gc::launch_pool_wait(&d_pool->done_total, published, nullptr, up_stream);
            (void)pool_stop(nullptr);
            (void)hipStreamSynchronize(up_stream);
            if (pool_stream) (void)hipStreamSynchronize(pool_stream);
        }
        for (hipStream_t st : streams) (void)hipStreamSynchronize(st);
    }
    void release() {
        drain();
        for (hipStream_t st : streams) (void)hipStreamDestroy(st);
        for (hipEvent_t ev : tail) (void)hipEventDestroy(ev);
        if (ctx_ev) (void)hipEventDestroy(ctx_ev);
        if (d_ver) (void)hipFree(d_ver);
        if (d_rd) (void)hipFree(d_rd);
        if (up_stream) (void)hipStreamDestroy(up_stream);
        if (pool_stream) (void)hipStreamDestroy(pool_stream);
        if (pool_ev) (void)hipEventDestroy(pool_ev);
        if (pool_go) (void)hipEventDestroy(pool_go);
        if (d_pool) (void)hipFree(d_pool);
        up_stream = pool_stream = nullptr, pool_ev = pool_go = nullptr, d_pool = nullptr, pool = persist = pool_running = false;
        streams.clear(), tail.clear(), used.clear();
        d_ver = d_rd = nullptr, ctx_ev = nullptr, cap = 0;
    }
};
// a pass of the ctx stream (a big step) has read in_idx[0, nin) and written out_idx[0, nout) (0xffffffff: not stored): the
// device counts follow (stream_group.cpp); the caller advances the host mirrors
void launch_df_bump(uint32_t *d_ver, uint32_t *d_rd, const uint32_t *d_in_idx, uint32_t nin, const uint32_t *d_out_idx, uint32_t nout,
                    hipStream_t st);

struct GroupWindow {
    std::deque<uint32_t> open;      // slots of the open groups, oldest first
    uint32_t first_seq = 1;         // sequence number of open.front()
    // per wire: sequence number of the latest group that reads / writes it (stale if < first_seq); chain fusion: which launch
    // unit of that group writes it (wrj) as which output of which of the group's steps (wrm = step << 20 | output index), and
    // which unit reads it (rdj; kFuseMulti: several)
    struct WireRec {
        uint32_t wr = 0, rd = 0, wrj = 0, rdj = 0, wrm = 0;
    };
    std::vector<WireRec> rec;
    void ensure(size_t n) {
        if (rec.size() < n) rec.resize(n);
    }
    // the records of these wires on their way into the cache (a step names a few hundred wires spread over a table of tens of
    // megabytes: the look-ups below are a chain of cache misses unless the loads are in flight together)
    void prefetch(const uint32_t *wires, uint32_t n) const {
        for (uint32_t i = 0; i < n; i += 3)  // (consecutive ids share cache lines: every third record is enough)
            if (wires[i] < rec.size()) __builtin_prefetch(&rec[wires[i]]);
    }
    // index into `open` of the earliest group the step may join (== open.size(): it needs a new group)
    uint32_t place(const uint32_t *reads, uint32_t nr, const uint32_t *writes, uint32_t nw) const {
        uint32_t lo = first_seq;
        for (uint32_t i = 0; i < nr; i++) {
            const uint32_t w = rec[reads[i]].wr;
            if (w >= lo) lo = w + 1;  // read after write
        }
        for (uint32_t j = 0; j < nw; j++) {
            if (writes[j] == 0xffffffffu) continue;
            const uint32_t w = rec[writes[j]].wr, r = rec[writes[j]].rd;
            if (w >= lo) lo = w + 1;  // write after write
            if (r >= lo) lo = r + 1;  // write after read
        }
        return lo - first_seq;
    }
    // place(), and chain fusion's question in the same walk: do ALL the step's conflicts with the latest open group it has any
    // with (the group before the one place() names) sit in ONE launch unit?  *unit = that unit, kFuseMulti if there are
    // several, kFuseNone if the step conflicts with nothing in the window.
    uint32_t place_fuse(const uint32_t *reads, uint32_t nr, const uint32_t *writes, uint32_t nw, uint32_t *unit) const {
        uint32_t best = 0, bj = kFuseNone;
        auto consider = [&](uint32_t seq, uint32_t job) {
            if (seq < first_seq) return;
            if (seq > best) best = seq, bj = job;
            else if (seq == best && job != bj) bj = kFuseMulti;
        };
        for (uint32_t i = 0; i < nr; i++) {
            const WireRec &r = rec[reads[i]];
            consider(r.wr, r.wrj);
        }
        for (uint32_t j = 0; j < nw; j++) {
            if (writes[j] == 0xffffffffu) continue;
            const WireRec &r = rec[writes[j]];
            consider(r.wr, r.wrj);
            consider(r.rd, r.rdj);
        }
        *unit = bj;
        return best ? best + 1 - first_seq : 0;
    }
    // The units of the open group with sequence number `seq` that the step conflicts with (distinct, at most `max`), for a
    // step that joins that group as a unit that waits for them.  Returns their number, or kDepsAllEarlier where the records
    // cannot name them (a wire several units read; more than `max`).
    uint32_t conflict_units(const uint32_t *reads, uint32_t nr, const uint32_t *writes, uint32_t nw, uint32_t seq, uint32_t *units,
                            uint32_t max) const {
        uint32_t n = 0;
        bool all = false;
        auto add = [&](uint32_t u) {
            if (u >= kFuseMulti) {
                all = true;
                return;
            }
            for (uint32_t i = 0; i < n; i++)
                if (units[i] == u) return;
            if (n < max) units[n++] = u;
            else all = true;
        };
        for (uint32_t i = 0; i < nr && !all; i++) {
            const WireRec &r = rec[reads[i]];
            if (r.wr == seq) add(r.wrj);
        }
        for (uint32_t j = 0; j < nw && !all; j++) {
            if (writes[j] == 0xffffffffu) continue;
            const WireRec &r = rec[writes[j]];
            if (r.wr == seq) add(r.wrj);
            if (r.rd == seq) add(r.rdj);
        }
        return all ? kDepsAllEarlier : n;
    }
    // unit / step: the launch unit of the group that the step is (part of), and its index among the group's steps
    void mark(uint32_t index, const uint32_t *reads, uint32_t nr, const uint32_t *writes, uint32_t nw, uint32_t unit = 0,
              uint32_t step = 0) {
        const uint32_t seq = first_seq + index;
        for (uint32_t i = 0; i < nr; i++) {
            WireRec &r = rec[reads[i]];
            if (r.rd < seq) r.rd = seq, r.rdj = unit;
            else if (r.rd == seq && r.rdj != unit) r.rdj = kFuseMulti;
        }
        for (uint32_t j = 0; j < nw; j++)
            if (writes[j] != 0xffffffffu) {
                WireRec &r = rec[writes[j]];
                r.wr = seq, r.wrj = unit, r.wrm = (step << 20) | j;
                // (who read the wire before this write is of no interest any more: whoever conflicts with them conflicts
                // with this writer, which is behind them)
                r.rd = 0, r.rdj = 0;
            }
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
        for (uint32_t i = 0; i < nr; i++) q = std::max(q, rec[reads[i]].wr);
        for (uint32_t j = 0; j < nw; j++)
            if (writes[j] != 0xffffffffu) q = std::max(q, std::max(rec[writes[j]].wr, rec[writes[j]].rd));
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
            std::fill(rec.begin(), rec.end(), WireRec{});
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
constexpr size_t kKeepQueued = 2;  // (3 / 4 / 6 / 8 measured on the Ed25519-shaped program, round 5: inside the run-to-run spread)
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
    static void setup_ctx(gc_ctx *ctx);  // stream_lanes.cpp
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

// ---- device-side serialiser (stream_serialise.cpp; stream_garble.go:391-446) ----------------------------------------------
constexpr uint32_t kSerThreads = 256, kSerPer = 4, kSerGates = kSerThreads * kSerPer;

struct SerArgs {
    const uint32_t *gw;   // {in0, in1, out} per gate
    const uint8_t *ops;
    const uint32_t *row_of_gate;
    const uint32_t *in, *out;  // wire maps of this call
    uint32_t ngates, first_tmp, first_out;
};

// ---- the serialiser of a step group (garbler): workgroup (p, j) = piece p of job j ------------------------------------
// The job's gates in the wire format (stream_garble.go:391-446), gate order, into the job's byte slot; the byte count
// into *size_out.  Runs on the copy stream behind the group's garbling kernel, beside the NEXT group's garbling (the
// output labels went back into the wire store in the garbling kernel's own epilogue).
struct FinJob {
    SerArgs a;
    const uint4 *T;
    uint8_t *bytes;
    uint32_t *size_out;
};

// a big step / a deep step of many gates, spread over the chip: byte size of every block of kSerGates gates and their
// exclusive scan into boff[0 .. nblocks] (boff[nblocks] = total, also into *total_out when given), then every gate to its
// byte offset (T: the table rows, lt their layout)
void ser_sizes_scan(const SerArgs &a, uint64_t *boff, uint32_t nblocks, uint32_t *total_out, hipStream_t s);
void ser_write(const SerArgs &a, const uint64_t *boff, uint32_t nblocks, const uint4 *T, const Layout &lt, uint8_t *buf, hipStream_t s);
// a step group: job j into its byte slot, in pieces of 512 gates — one workgroup each (d_jobs: device array of n records;
// max_gates: the gate count of the largest job)
void ser_group(const FinJob *d_jobs, uint32_t n, uint32_t max_gates, hipStream_t s);

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
uint64_t circuit_hash(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t nin, uint32_t nout);

// ---- circuit cache, launch slots, launch sequence of a group (stream_group.cpp) -------------------------------------------
CircEntry *cache_find(CircCache &cache, uint64_t h, const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t nin,
                      uint32_t nout);
CircEntry *cache_find_keys(CircCache &cache, uint64_t h, const std::vector<CircKey> &keys, uint32_t nwires, uint32_t nin,
                           uint32_t nout);
CircEntry *cache_put_keys(CircCache &cache, uint64_t h, gc_circ *circ, const std::vector<CircKey> &keys, uint32_t nwires,
                          uint32_t nin, uint32_t nout);
CircEntry *cache_put(CircCache &cache, uint64_t h, gc_circ *circ, const gc_gate *gates, uint32_t ngates, uint32_t nwires,
                     uint32_t nin, uint32_t nout);
size_t cache_budget_from_env();
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

// ---- chain fusion (stream_fuse.cpp) ------------------------------------------------------------------------------------------
// The merged plan of a chain of steps: their gate lists concatenated with the wires re-named — an input that an earlier step
// of the chain produces IS that step's output wire; the others become the merged circuit's inputs, in order; every step's
// outputs are outputs of the merged circuit — the hash tweak starting over at every step (stream_garble.go:174) and the table
// rows counted through (step k's rows: [row_base[k], row_base[k + 1])).  Cached per ctx by (circuits, wiring): a compiled
// program repeats its chains (the ten sums and the carry chain of a field multiplication) thousands of times.
struct FusedPlan {
    gc_circ *circ = nullptr;     // the merged circuit (owned by the ctx's cache; null: the chain has no one-workgroup plan)
    gc::FlatJob job{};           // its circuit-constant job record, LDS need, OR gates
    size_t lds = 0;
    bool has_or = false;
    uint32_t n_ext = 0, n_out = 0, n_steps = 0;  // merged inputs / outputs; barriers of its pass
    std::vector<uint32_t> gate_base, row_base;   // per step (+ the totals at the end)
    std::vector<uint32_t> key;                   // what it was built from (compared on a hit)
};
// one step of a chain as merge_chain sees it: its gates in its own wire ids (inputs [0, nin), outputs the last nout wires) and
// the source of every input (null: all from the wire store)
struct ChainStep {
    const gc_gate *gates;
    uint32_t ngates, nwires, nin, nout;
    const uint32_t *wiring;
};
// the chain as ONE gate list: wires [0, n_ext) the inputs read from the wire store (in step order), then every step's
// private wires, then every step's outputs (in step order: the merged circuit's outputs); gate_base[k] = first gate of step k
// (+ the total): where the hash tweak starts over
int merge_chain(const ChainStep *steps, uint32_t n, std::vector<gc_gate> *all, std::vector<uint32_t> *gate_base, uint32_t *n_ext,
                uint32_t *n_tmp, uint32_t *n_out);
struct FuseStats {
    uint64_t units = 0, steps = 0;  // launch units of several steps, and the steps in them
    uint64_t appended = 0;          // steps appended to a unit when they were queued
    uint64_t unplanned = 0;         // units that ran step by step because their chain was met for the first time (or the cache is full)
    uint64_t built = 0, unfit = 0;  // merged plans this stream had to build; units that ran step by step (no one-workgroup plan)
    uint64_t waiting = 0;           // units that joined the group they conflict with and wait, on the device, for units of it
};
struct FuseMember {
    const CircEntry *ent;
    const uint32_t *wiring;  // per input: kFuseNone (from the wire store) or member << 24 | output index; null: all from the store
};
// the merged plan of the chain (built, by the ctx's planner thread, when the chain has been met a few times; plan->circ == nullptr when it does not fit
// one workgroup); nullptr: not planned (yet) / no memory / the cache is full — the caller then runs the steps one after the
// other.  *built: this call planned it.
const FusedPlan *fuse_plan(gc_ctx *ctx, bool eval_form, const FuseMember *members, uint32_t n, bool *built);
// the ctx-wide identity of a circuit's content (what the merged plans are keyed by): 0 when the registry is full or the
// circuit too big to be part of a chain
uint32_t fuse_register(gc_ctx *ctx, bool eval_form, uint64_t hash, const std::vector<CircKey> &gates, uint32_t nwires, uint32_t nin,
                       uint32_t nout);
// chain fusion is on (GC_STREAM_NO_FUSE switches it off: every step its own launch unit, as before round 5)
bool fuse_enabled();
// the step joins unit `unit` of group g (its index there is jobs.size(): the caller pushes the JobRec next); n_ext: how many
// of its inputs come from the wire store
void wg_append(Slot &g, uint32_t unit, JobRec *j, const CircEntry *ent, uint32_t n_ext, uint64_t shape);
// The running hash of a unit's shape when a step with this circuit and wiring joins it, and what earlier chains of that shape
// say about its DEPTH (dependent hash phases of the merged plan up to and including that step; 0: not known yet).  A unit
// is one workgroup: a chain that would run for a thousand phases holds up its whole group, so the caller stops appending
// where the hint exceeds kFuseDepth — the first chain of a shape runs as long as the other caps allow, and tells.
uint64_t fuse_shape(uint64_t shape, const CircEntry *ent, const uint32_t *wiring);
uint32_t fuse_depth_hint(gc_ctx *ctx, uint64_t shape);
// a new launch unit for a step about to be pushed to g.jobs; returns its index
uint32_t wg_new(Slot &g, JobRec *j, const CircEntry *ent, bool may_fuse);
// Units that wait inside a launch.  wg_covering: the step conflicts with these units of group g (n of them, known); the one
// it could be APPENDED to — the latest, if every other one is done by the time that unit starts (WgRec::behind) — or
// kFuseMulti.  wg_wait: unit `unit` (new, the group's last) waits for these units (n == kDepsAllEarlier: for every earlier one).
uint32_t wg_covering(const Slot &g, const uint32_t *units, uint32_t n);
void wg_wait(Slot &g, uint32_t unit, const uint32_t *units, uint32_t n);

bool entry_is_small(CircEntry *e);
bool entry_is_deep(CircEntry *e, uint32_t min_steps, bool in_stream);
Slot *slot_new(gc_ctx *ctx, std::vector<std::unique_ptr<Slot>> &slots, uint32_t *index, bool big = false);
void deep_after(const GroupWindow &win, const std::vector<std::unique_ptr<Slot>> &slots, uint32_t cs, Slot *ng);
int launch_group(gc_ctx *ctx, Slot &g, bool eval, DevStore &store, const uint32_t *d_rk, const uint4 *d_R, int rounds,
                 hipStream_t copy_stream, DeepLanes &deep, FuseStats *fstats, bool one_stream = false, Dataflow *df = nullptr);

}  // namespace gcs
