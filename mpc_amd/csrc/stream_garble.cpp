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
#include "stream_internal.h"

using namespace gcs;

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
    std::vector<uint32_t> wiring_scratch;      // chain fusion: the input sources of the step being queued
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
    CopyPool copier;              // gc_stream_garble_finish_async: the copies into the caller's buffer, off this thread
    FuseStats fuse;               // chain fusion: launch units of several steps, merged plans built
    bool use_deps = deps_wanted();  // units that wait inside a launch (stream_internal.h: kUnitDeps)
    Dataflow df;                    // units ordered by the versions of their wires, across launches (GC_STREAM_DATAFLOW)
    StageProf prof;
    uint64_t n_steps_total = 0;
    // gc_stream_garble_finish_view: the slot whose pinned bytes the caller is still reading (given back by the next finish),
    // and pinned staging for the bytes of a big step
    uint32_t view_slot = 0xffffffffu;
    uint8_t *view_buf = nullptr;
    size_t view_cap = 0;
};

namespace {

inline uint64_t be64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}
inline void ensure(gc_stream *s, uint32_t max) {  // ensureWires, 64 Ki-wire pages (:95-100)
    if (max < s->store.host.size()) return;
    s->store.ensure(((size_t)max / 0x10000 + 1) * 0x10000);
}

// the oldest open group leaves the window and is launched
int launch_oldest(gc_stream *s, bool one_stream = false) {
    if (s->win.open.empty()) return GC_OK;
    const uint32_t seq = s->win.first_seq, slot = s->win.pop();
    Slot &g = *s->slots[slot];
    s->n_groups++;
    s->n_group_steps += g.jobs.size();
    s->prof.lap(StageProf::kOther);
    const int rc = launch_group(s->ctx, g, false, s->store, s->d_rk, s->d_R, s->rounds, s->copy_stream, s->deep, &s->fuse, one_stream,
                                &s->df);
    s->prof.lap(StageProf::kLaunch);
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
            ent->uid = fuse_register(s->ctx, false, h, ent->gates, nwires, nin, nout);  // (chain fusion: its ctx-wide identity)
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
        if (e == hipSuccess && dataflow_wanted() && fuse_enabled()) {  // (experiment: off unless GC_STREAM_DATAFLOW=1)
            e = s->df.setup();
            s->df.on = e == hipSuccess;
            // (GC_STREAM_DEPS=1 on top: steps that conflict with several units of an open group join it — units that depend on
            // one another inside a launch, in ticket order — instead of opening a later group)
        }
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
    s->copier.wait();  // (copies out of the slots' pinned bytes that are still under way)
    s->copier.stop();
    s->prof.print("garbler", s->n_steps_total);
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
        if (GroupTimeline::enabled()) group_timeline().print_and_clear();
    }
    s->deep.release();
    s->df.release();
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

int gc_stream_fuse_stats(const gc_stream *s, uint64_t *fused_units, uint64_t *fused_steps, uint64_t *plans_built, uint64_t *unfit) {
    if (!s) return GC_E_ARG;
    if (fused_units) *fused_units = s->fuse.units;
    if (fused_steps) *fused_steps = s->fuse.steps;
    if (plans_built) *plans_built = s->fuse.built;
    if (unfit) *unfit = s->fuse.unfit;
    return GC_OK;
}

int gc_stream_wait_stats(const gc_stream *s, uint64_t *waiting_units) {
    if (!s) return GC_E_ARG;
    if (waiting_units) *waiting_units = s->fuse.waiting;
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
    if (s->df.on) s->df.drain();              // ... or a group on one of the rotating streams (Dataflow)
    gc_label l0;
    rc = s->store.get(s->ctx, w, &l0);
    if (rc != GC_OK) return rc;
    if ((rc = gc_ctx_coop_check(s->ctx)) != GC_OK) return rc;  // (the stream was waited for: a pass that lost a workgroup is noted here)
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
    s->prof.start();
    s->n_steps_total++;
    if (s->queue.size() >= kMaxPending) return GC_E_ARG;
    // in[] and out[] may overlap (a circuit whose last wires are input wires): initCircuit (:102-114) takes both as they
    // are, Get / Set resolve a wire through in[] first (:131-157), so such an output id is simply never written
    if (nin > nwires || nout > nwires) return GC_E_ARG;
    const uint32_t first_tmp = nin, first_out = nwires - nout;
    // initCircuit (:102-114)
    uint32_t mx = 0, in_hi = 0, out_lo = 0xffffffffu, out_hi = 0;
    for (uint32_t i = 0; i < nin; i++) in_hi = std::max(in_hi, in[i]);
    for (uint32_t i = 0; i < nout; i++) out_lo = std::min(out_lo, out[i]), out_hi = std::max(out_hi, out[i]);
    mx = std::max(in_hi, out_hi);
    ensure(s, mx);
    // (in[] and out[] can only name a common wire if an input id lies inside the id range of the outputs: as a rule — results
    // take fresh ids from the allocator, operands and constants lie elsewhere — none does, and the per-wire look of the
    // aliasing check below, two passes over tables the size of the wire store, is skipped)
    bool ranges_overlap = false;
    for (uint32_t i = 0; i < nin && nout; i++) ranges_overlap |= in[i] >= out_lo && in[i] <= out_hi;
    gc_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    GC_HIP(hipSetDevice(ctx->device));
    StreamTrace tr;

    // in[] / out[] naming the same GLOBAL wire (wire-id re-use, in-place update): the reference resolves
    // stream.wire(index) per gate (:131-157), so a gate that reads the input-mapped wire after the gate that Set the
    // output-mapped one sees the NEW label.  The device garbles from a snapshot of the inputs: redirect such reads to
    // the producing circuit wire (same global id and flags on the wire, so the serialised bytes do not change).
    bool was_aliased = false;
    if (ngates && ranges_overlap) {
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
        was_aliased = aliased;
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

    if (ngates && s->win.rec.size() >= s->store.host.size()) {
        s->win.prefetch(in, nin);
        s->win.prefetch(out, nout);
    }
    s->prof.lap(StageProf::kGuess);  // (alias check, look-up, skip marks)
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
            if (s->df.on) GC_HIP(s->df.join(ctx->stream));  // (the upload overwrites store entries: behind every group launched so far)
            if (s->df.persist && s->store.host.size() > s->store.cap) s->df.drain();  // (a growing store moves behind a device-wide wait)
            int rcs = s->store.flush(ctx);
            if (rcs != GC_OK) return rcs;
        }
        s->win.ensure(s->store.host.size());
        if (is_deep || s->deep.n_inflight) s->deep.ensure(s->store.host.size());
        const size_t wbytes = up256((size_t)ent->job.w_tile * 16) + up256((size_t)ent->job.t_tile * 16);
        // Chain fusion (stream_fuse.cpp): when everything the step conflicts with in the latest group it conflicts with at all
        // (group gi - 1) is ONE launch unit, a short step is appended to that unit — the chain runs as one planned job — instead
        // of waiting for that whole group in a later one.
        uint32_t unit = kFuseNone;
        const uint32_t *skip = s->skip_scratch.data();
        uint32_t gi = fuse_enabled() ? s->win.place_fuse(in, nin, skip, nout, &unit) : s->win.place(in, nin, skip, nout);
        const bool may_fuse = fuse_enabled() && !is_deep && !was_aliased && ent->uid != 0 && first_out >= first_tmp;
        // conflicts with SEVERAL units of that group: which ones (stream_internal.h: kUnitDeps) — a step can still be appended
        // to the latest of them if that one starts behind all the others, else it joins the group as a unit that waits
        uint32_t dep_units[kUnitDeps], ndeps = 0;
        const bool can_wait = fuse_enabled() && s->use_deps && !is_deep && gi > 0;
        if (can_wait && unit == kFuseMulti) {
            ndeps = s->win.conflict_units(in, nin, skip, nout, s->win.first_seq + gi - 1, dep_units, kUnitDeps);
            unit = wg_covering(*s->slots[s->win.open[gi - 1]], dep_units, ndeps);
        } else if (can_wait) {
            dep_units[0] = unit, ndeps = 1;
        }
        bool fuse = false;
        uint64_t shape = 0;
        uint32_t n_ext = 0;
        if (may_fuse && gi > 0 && unit < kFuseMulti && ngates <= kFuseTailGates) {
            const Slot &fg = *s->slots[s->win.open[gi - 1]];
            const WgRec &w = fg.wgs[unit];
            fuse = w.open && w.n < kFuseMembers && w.gates + ngates <= kFuseGates && w.slots + ent->job.zslot + 1 <= kFuseSlots &&
                   w.inputs + nin <= kFuseInputs && fg.jobs.size() < kGroupSteps &&
                   fg.arena_used + fg.down_used + wbytes + ent->ser_long <= kGroupBytes;
            if (fuse) {
                // where every input comes from: an earlier step of the unit (the window's record of the wire names it) or the
                // wire store; and is a chain of this shape known to run too long for one workgroup?
                const uint32_t seq = s->win.first_seq + gi - 1;
                s->wiring_scratch.resize(nin);
                for (uint32_t i = 0; i < nin; i++) {
                    const GroupWindow::WireRec &r = s->win.rec[in[i]];
                    if (r.wr == seq && r.wrj == unit) s->wiring_scratch[i] = (fg.jobs[r.wrm >> 20].member << 24) | (r.wrm & 0xfffffu);
                    else s->wiring_scratch[i] = kFuseNone, n_ext++;
                }
                shape = fuse_shape(w.shape, ent, s->wiring_scratch.data());
                const uint32_t hint = fuse_depth_hint(ctx, shape);
                fuse = hint ? hint <= fuse_depth_cap() : w.depth_sum + ent->circ->plan.p.n_hash_phases <= 2 * fuse_depth_cap();
            }
        }
        uint32_t slot_idx = 0;
        auto full = [&](const Slot &g) {
            return g.wgs.size() >= kGroupJobs || g.jobs.size() >= kGroupSteps ||
                   g.arena_used + g.down_used + wbytes + ent->ser_long > kGroupBytes;
        };
        // not fused: the step still joins the group it conflicts with, as a unit that waits on the device for the units it
        // conflicts with there (stream_internal.h: kUnitDeps) — where that group has room
        bool waits = false;
        if (!fuse && can_wait && !full(*s->slots[s->win.open[gi - 1]])) {
            waits = true;
            gi--;
        }
        s->prof.lap(StageProf::kPlace);
        if (fuse || waits) {
            if (fuse) gi--;
            slot_idx = s->win.open[gi];
        } else if (is_deep) {
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
        const uint32_t step_idx = (uint32_t)g.jobs.size();
        if (fuse) {
            // the wiring goes with the step; outputs of earlier steps of the unit that this one overwrites: their stores are
            // dropped (launch_group)
            const uint32_t seq = s->win.first_seq + gi;
            j.off_wiring = g.wiring.size();
            g.wiring.insert(g.wiring.end(), s->wiring_scratch.begin(), s->wiring_scratch.begin() + nin);
            for (uint32_t k = 0; k < nout; k++) {
                if (skip[k] == 0xffffffffu) continue;
                const GroupWindow::WireRec &r = s->win.rec[out[k]];
                if (r.wr == seq && r.wrj == unit) g.kills.emplace_back(r.wrm >> 20, r.wrm & 0xfffffu);
            }
            wg_append(g, unit, &j, ent, n_ext, shape);
            s->fuse.appended++;
        } else {
            unit = wg_new(g, &j, ent, may_fuse);
            if (waits) {
                wg_wait(g, unit, dep_units, ndeps);
                s->fuse.waiting++;
            }
        }
        g.jobs.push_back(j);
        s->prof.lap(StageProf::kQueue);
        if (!is_deep) {
            s->win.mark(gi, in, nin, s->skip_scratch.data(), nout, unit, step_idx);
            // a deep step in flight that this one must follow: the group waits for it (and for the older ones of its lane)
            if (s->deep.n_inflight) g.deps.merge(s->deep.conflicts(in, nin, s->skip_scratch.data(), nout));
        }
        if (is_deep) {  // launched at once, on its lane
            int rcl = launch_group(ctx, g, false, s->store, s->d_rk, s->d_R, s->rounds, s->copy_stream, s->deep, &s->fuse, false, &s->df);
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
        s->prof.lap(StageProf::kMark);
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
        if (s->df.on) {  // ... and behind every group on the rotating streams (Dataflow)
            std::lock_guard<std::mutex> lk(ctx->mu);
            GC_HIP(s->df.join(st));
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
    if (s->df.on) {
        // dataflow across launches: the pass has read in[] and written the stored outputs — the wires' counts follow, on the device
        // behind the pass and in the host's mirrors, and every later group launch waits for this point of the ctx stream
        std::lock_guard<std::mutex> lk(ctx->mu);
        GC_HIP(s->df.ensure(s->store.cap));
        launch_df_bump(s->df.d_ver, s->df.d_rd, b.d_io, nin, b.d_io + nin + nout, nout, st);
        GC_HIP(hipGetLastError());
        for (uint32_t i = 0; i < nin; i++) s->df.vr[in[i]].rd++;
        for (uint32_t j = 0; j < nout; j++)
            if (s->skip_scratch[j] != 0xffffffffu) s->df.vr[out[j]].ver++;
        GC_HIP(s->df.fence_from(st));
    }
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
            ser_sizes_scan(a, b.d_boff, nblocks, nullptr, ss);
            e = hipMemcpyAsync(b.need, b.d_boff + nblocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ss);
        }
        if (e == hipSuccess) {
            ser_write(a, b.d_boff, nblocks, bt->d_T, bt->g.lt, b.d_bytes, ss);
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
// slots whose steps have all been handed out and whose deferred copies are done go back to the free ones
static void retire_slots(gc_stream *s) {
    for (auto &sl : s->slots)
        if (sl->retire && sl->copies.load(std::memory_order_acquire) == 0) {
            if (sl->deep_id) s->deep.retire(sl->lane, sl->deep_id);
            sl->reset();
        }
}

static int stream_finish(gc_stream *s, uint8_t *buf, size_t cap, size_t *written, const uint8_t **view, bool async = false) {
    if (!s || (!buf && !view) || !written || s->queue.empty()) return GC_E_ARG;
    s->prof.start();
    StreamTrace tr;
    if (s->view_slot != 0xffffffffu) {  // the slot whose bytes the last view pointed into
        Slot &v = *s->slots[s->view_slot];
        // (earlier steps of the same group may have been handed out by finish_async: their copies read v.h_down until the
        // copier threads are through — such a slot is given back by retire_slots, never reset under them)
        if (v.copies.load(std::memory_order_acquire) != 0) {
            v.retire = true;
        } else {
            if (v.deep_id) s->deep.retire(v.lane, v.deep_id);
            v.reset();
        }
        s->view_slot = 0xffffffffu;
    }
    retire_slots(s);  // (every kind of finish: a slot marked `retire` is not kept until the next copies_wait)
    const StepRef ref = s->queue.front();
    Slot &g = *s->slots[ref.slot];
    while (g.kind == Slot::kGroup && !g.launched) {  // the oldest step sits in an open group: launch up to that one
        // (the only step queued, in the only open group: the caller garbles one instruction at a time — launch_group)
        int rc = launch_oldest(s, s->queue.size() == 1 && s->win.open.size() == 1);
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
        // (Also when this group's bytes are there already: a group that the full window sent to the GPU early would otherwise
        // leave the GPU with nothing behind it until the caller reaches the NEXT group — and sits through that one's whole
        // kernel, serialiser and copy.)
        while (!s->win.open.empty() && s->ctxq.hungry(s->slots)) {
            int rcq = launch_oldest(s);
            if (rcq != GC_OK) break;
        }
        (void)hipGetLastError();
        s->prof.lap(StageProf::kOther);
        GroupTimeline::G *tl = GroupTimeline::enabled() && g.kind == Slot::kGroup ? group_timeline().find(g.launch_no) : nullptr;
        if (tl) tl->h_w0 = group_timeline().now();
        hipError_t e = hipEventSynchronize(g.done);
        if (tl) tl->h_w1 = group_timeline().now();
        s->prof.lap(StageProf::kWait);
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
            else if (need && async) s->copier.submit(buf, g.h_down + sizes_bytes + j.off_bytes, need, &g.copies);
            else if (need) std::memcpy(buf, g.h_down + sizes_bytes + j.off_bytes, need);
        }
        if (++g.handed == g.jobs.size()) {
            if (view && rc == GC_OK) {
                s->view_slot = ref.slot;  // (given back by the next finish: the caller still reads its bytes)
            } else if (g.copies.load(std::memory_order_acquire) != 0) {
                g.retire = true;          // (given back when its copies are done: retire_slots)
            } else {
                if (g.deep_id) s->deep.retire(g.lane, g.deep_id);  // (its kernel has run: `done` sits behind it)
                g.reset();
            }
        }
        s->prof.lap(StageProf::kFinish);
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

int gc_stream_garble_finish_async(gc_stream *s, uint8_t *buf, size_t cap, size_t *written) try {
    if (!s || !buf) return GC_E_ARG;
    return stream_finish(s, buf, cap, written, nullptr, true);
} catch (...) {
    return gc::on_exception();
}

int gc_stream_garble_copies_wait(gc_stream *s) try {
    if (!s) return GC_E_ARG;
    s->copier.wait();
    retire_slots(s);
    return GC_OK;
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

}  // extern "C"
