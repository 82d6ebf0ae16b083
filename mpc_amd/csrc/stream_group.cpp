// stream_group.cpp — shared by the streaming garbler and evaluator: the per-stream circuit cache, which circuits one
// workgroup can run (small / deep), launch slots, and the launch sequence of a step group (see stream_garble.cpp's head).
#include "stream_internal.h"

namespace gcs {

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
Slot *slot_new(gc_ctx *ctx, std::vector<std::unique_ptr<Slot>> &slots, uint32_t *index, bool big) {
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

// table rows of the evaluator's fused chains: every step's rows, wherever the parser put them in the group's upload region, into
// the chain's one table array (workgroup c = copy c)
namespace {
struct RowCopy {
    const uint4 *src;
    uint4 *dst;
    uint32_t n, pad_;
};
// dataflow across launches: a pass that is no group launch (a big step on the ctx stream) has read / written these wires
__global__ __launch_bounds__(256) void k_df_bump(uint32_t *ver, uint32_t *rd, const uint32_t *in_idx, uint32_t nin, const uint32_t *out_idx,
                                                 uint32_t nout) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nin) {
        __hip_atomic_fetch_add(rd + in_idx[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (i < nin + nout) {
        const uint32_t w = out_idx[i - nin];
        if (w != 0xffffffffu) __hip_atomic_fetch_add(ver + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void k_rows_copy(const RowCopy *rc) {
    const RowCopy c = rc[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < c.n; i += 256) c.dst[i] = c.src[i];
}
// ... and the rows of blocks the DEVICE matched (stream_eval_dev.cpp): still in the peer's bytes, in device memory — row r of
// the block is the 16 bytes BE(D0) || BE(D1) at src + row_off[r] (any alignment), its label in host order goes to dst[r]
struct RowGather {
    const uint8_t *src;
    const uint32_t *row_off;
    uint4 *dst;
    uint32_t n, pad_;
};
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
__global__ __launch_bounds__(256) void k_rows_gather(const RowGather *rg) {
    const RowGather c = rg[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < c.n; i += 256) {
        const u32_unaligned *p = (const u32_unaligned *)(c.src + c.row_off[i]);
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
        // uint4 label: (x, y) = D0 low / high, (z, w) = D1 low / high
        c.dst[i] = make_uint4(__builtin_bswap32(w1), __builtin_bswap32(w0), __builtin_bswap32(w3), __builtin_bswap32(w2));
    }
}
}  // namespace

GroupTimeline &group_timeline() {
    static GroupTimeline t;
    return t;
}
void GroupTimeline::print_and_clear() {
    for (const G &g : groups) {
        float a = 0, b = 0, c = 0;
        if (g.d) (void)hipEventSynchronize(g.d);
        if (ref && g.k0) (void)hipEventElapsedTime(&a, ref, g.k0);
        if (ref && g.k1) (void)hipEventElapsedTime(&b, ref, g.k1);
        if (ref && g.d) (void)hipEventElapsedTime(&c, ref, g.d);
        std::fprintf(stderr, "[gc timeline] group %llu steps %u host launch %.1f %.1f wait %.1f %.1f gpu head %.1f kernel_end %.1f bytes %.1f\n",
                     (unsigned long long)g.launch_no, g.steps, g.h_l0, g.h_l1, g.h_w0, g.h_w1, a * 1e3, b * 1e3, c * 1e3);
        if (g.k0) (void)hipEventDestroy(g.k0);
        if (g.k1) (void)hipEventDestroy(g.k1);
        if (g.d) (void)hipEventDestroy(g.d);
    }
    groups.clear();
    if (ref) (void)hipEventDestroy(ref);
    ref = nullptr;
}

// Launch sequence of a group (see the head of stream_garble.cpp).  eval: the jobs' table rows are part of the upload region and
// nothing comes back.  On return the slot is `launched`; a failure is kept in slot.error for the group's steps.
// A deep step (g.deep_id != 0, one job) takes the same sequence on its lane, behind an event recorded on the ctx stream here
// and behind the deep steps of the other lanes named by g.deps; a group of small steps on the ctx stream waits for the deep
// steps of g.deps (DeepLanes).
// Chain fusion (stream_fuse.cpp): a launch unit of several steps runs as ONE job on its merged plan — inputs gathered from the
// wire store for the whole chain, every step's outputs scattered back at the end (an output that a later step of the chain
// writes again is dropped), the garbler's serialiser finding step k's rows in the chain's table array at row_base[k].  A chain
// without a merged plan (not met often enough yet, or none fits a workgroup) still is ONE workgroup: it runs its steps' own
// jobs one after the other, through the wire store (fused_flat_kernels.hip: k_*_flat_jobs).
// one_stream (garbler): the caller waits for exactly this group and nothing else is queued (the unchanged caller: one
// Streaming.Garble at a time) — serialiser and bytes follow the kernel on ITS stream: nothing to run beside, and the hand-over
// to the copy stream (an event between two hardware queues) is latency the caller would sit through.
// (GC_STREAM_DATAFLOW=2: the lanes of the deep steps keep their events towards the groups — for comparison)
static bool df_free_lanes() {
    static const bool v = [] {
        const char *e = std::getenv("GC_STREAM_DATAFLOW");
        return !(e && e[0] == '2');
    }();
    return v;
}

void launch_df_bump(uint32_t *d_ver, uint32_t *d_rd, const uint32_t *d_in_idx, uint32_t nin, const uint32_t *d_out_idx, uint32_t nout,
                    hipStream_t st) {
    if (nin + nout == 0) return;
    hipLaunchKernelGGL(k_df_bump, dim3((nin + nout + 255) / 256), dim3(256), 0, st, d_ver, d_rd, d_in_idx, nin, d_out_idx, nout);
}

int launch_group(gc_ctx *ctx, Slot &g, bool eval, DevStore &store, const uint32_t *d_rk, const uint4 *d_R, int rounds,
                 hipStream_t copy_stream, DeepLanes &deep, FuseStats *fstats, bool one_stream, Dataflow *df) {
    const bool on_lane = g.deep_id != 0;
    if (df && (!df->on || eval)) df = nullptr;
    // dataflow across launches (stream_internal.h: Dataflow): the group goes to the next of the rotating streams — nothing orders
    // it against the groups before it but the versions of the wires its units read and write
    uint32_t df_k = 0;
    hipStream_t st = on_lane ? deep.lanes[(size_t)g.lane] : df ? df->next(&df_k) : ctx->stream;
    if (one_stream && !on_lane && !eval) copy_stream = st;
    const uint32_t n = (uint32_t)g.jobs.size(), nwg = (uint32_t)g.wgs.size();
    // ---- the merged plans of the chains (before the ctx lock: a chain met for the first time is planned here)
    std::vector<const FusedPlan *> plans(nwg, nullptr);
    uint32_t nrec = 0, ncopies = 0, ngathers = 0;
    if (eval)
        for (const JobRec &j : g.jobs) ngathers += j.d_block != nullptr && j.nrows != 0;
    size_t extra_up = 0, extra_arena = 0;
    size_t lds = g.lds;
    bool has_or = g.has_or;
    {
        std::vector<FuseMember> mem;
        for (uint32_t u = 0; u < nwg; u++) {
            const WgRec &w = g.wgs[u];
            if (w.n < 2) {
                nrec++;
                continue;
            }
            mem.clear();
            for (int32_t k = (int32_t)w.head; k >= 0; k = g.jobs[(size_t)k].next)
                mem.push_back(FuseMember{g.jobs[(size_t)k].ent, mem.empty() ? nullptr : g.wiring.data() + g.jobs[(size_t)k].off_wiring});
            bool built = false;
            const FusedPlan *fp = fuse_plan(ctx, eval, mem.data(), (uint32_t)mem.size(), &built);
            if (fstats) {
                fstats->units++;
                fstats->steps += w.n;
                fstats->built += built;
            }
            if (fp && fp->circ) {
                plans[u] = fp;
                nrec++;
                extra_up += up16(((size_t)fp->n_ext + fp->n_out) * sizeof(uint32_t));
                extra_arena += up256(fp->job.w_tile * 16) + up256(fp->job.t_tile * 16);
                if (eval) ncopies += w.n;
                lds = std::max(lds, fp->lds);
                has_or = has_or || fp->has_or;
            } else {
                nrec += w.n;
                if (fstats) (fp ? fstats->unfit : fstats->unplanned)++;
            }
        }
    }
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
    // (dataflow: an upload of host-set labels overwrites store entries on the ctx stream — behind every group launched so far)
    if (df && !store.dirty.empty() && (e = df->join(ctx->stream)) != hipSuccess) return fail("launch_group (dataflow join)", e);
    // (persistent workgroups: a store that has to GROW moves behind a device-wide wait — they must have left before it)
    if (df && df->persist && store.host.size() > store.cap) df->drain();
    int rcs = store.flush(ctx);  // host-set labels go up first; the store may move (its pointer is taken below)
    if (rcs != GC_OK) return g.error = rcs;
    if (df) {
        if ((e = df->ensure(store.cap)) != hipSuccess) return fail("launch_group (dataflow counts)", e);
        // what the ctx stream did to the store outside group launches (uploads, big steps) comes first
        if (!on_lane && store.up_ev) e = hipStreamWaitEvent(st, store.up_ev, 0);
        if (e == hipSuccess && df->ctx_ev_set) e = hipStreamWaitEvent(st, df->ctx_ev, 0);
        if (e != hipSuccess) return fail("launch_group (dataflow order)", e);
        // bounded: every unit launched must be able to become resident while it waits for another
        // (not under out-of-order issue: a unit is claimed by a workgroup that IS resident, in program order)
        while (!df->pool && !df->inflight.empty()) {
            const Dataflow::InFlight &f = df->inflight.front();
            const bool gone = !f.slot->launched || f.slot->launch_no != f.launch_no;
            if (!gone && !df->fits(nwg)) {
                (void)hipEventSynchronize(f.slot->kernel_ev);
                df->n_cap_waits++;
            } else if (!gone && hipEventQuery(f.slot->kernel_ev) != hipSuccess) {
                break;
            }
            df->account(f.units, false);
            df->inflight.pop_front();
        }
        (void)hipGetLastError();  // hipErrorNotReady
    }
    if (df && df_free_lanes()) {
        // dataflow: the versions of the wires order lanes and groups against one another as well — no events between them
        if (on_lane && store.up_ev) e = hipStreamWaitEvent(st, store.up_ev, 0);
    } else if (on_lane) {
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
    // upload region: [what the steps queued] [merged wire maps of the chains] [job records] [serialiser records] [row copies]
    // [first job record of every unit]
    const size_t off_maps = up16(g.up_used), off_fj = off_maps + extra_up, off_fin = off_fj + (size_t)nrec * sizeof(FlatJob);
    const size_t off_rc = off_fin + (size_t)n * sizeof(FinJob);
    const size_t off_rg = off_rc + (size_t)ncopies * sizeof(RowCopy);
    const size_t off_first = off_rg + (size_t)ngathers * sizeof(RowGather);
    // [ticket, unit count, error word, done-flags, the lists of the units that wait] (kernels.h: d_sync) where units of the
    // group wait for one another
    size_t sync_words = 0;
    if (g.has_waits) {
        sync_words = kSyncHead + (size_t)nwg + nwg + 1;
        for (uint32_t u = 0; u < nwg; u++) sync_words += g.wgs[u].dep_n == kDepsAllEarlier ? u : g.wgs[u].dep_n;
    }
    const size_t off_sync = off_first + up16((size_t)nwg * sizeof(uint32_t));
    // [the units' dataflow blocks: header, version per input, {prev, reads, next} per output] (an upper bound: every step a record)
    const size_t off_df = off_sync + up16(sync_words * sizeof(uint32_t));
    size_t df_bytes = 0;
    if (df)
        for (const JobRec &j : g.jobs) df_bytes += up16(sizeof(DfBlock) + ((size_t)j.nin + 3 * (size_t)j.nout) * sizeof(uint32_t));
    // [out-of-order issue: the units' ring entries]
    const bool pool = df && df->pool;
    const size_t off_pool = off_df + df_bytes;
    const size_t total_up = off_pool + (pool ? up16((size_t)nwg * sizeof(PoolEntry)) : 0);
    const size_t sizes_bytes = up256((size_t)n * sizeof(uint32_t));
    const size_t arena_chain = up256(g.arena_used);
    if ((e = g.reserve_up(total_up - g.up_used)) != hipSuccess) return fail("launch_group (pinned)", e);
    if ((e = grow_dev(ctx, &g.d_up, &g.d_up_cap, total_up)) != hipSuccess) return fail("launch_group (upload)", e);
    if ((e = grow_dev(ctx, &g.d_arena, &g.arena_cap, std::max<size_t>(arena_chain + extra_arena, 256))) != hipSuccess) return fail("launch_group (arena)", e);
    if (!eval) {
        if ((e = grow_dev(ctx, &g.d_down, &g.d_down_cap, sizes_bytes + g.down_used)) != hipSuccess) return fail("launch_group (bytes)", e);
        if ((e = grow_pin(ctx, &g.h_down, &g.h_down_cap, sizes_bytes + g.down_used)) != hipSuccess) return fail("launch_group (pinned bytes)", e);
    }
    FlatJob *fj = (FlatJob *)(g.h_up + off_fj);
    FinJob *fin = (FinJob *)(g.h_up + off_fin);
    RowCopy *rcp = (RowCopy *)(g.h_up + off_rc);
    RowGather *rgp = (RowGather *)(g.h_up + off_rg);
    uint32_t ngt = 0;
    uint32_t *first = (uint32_t *)(g.h_up + off_first);
    uint32_t rec = 0;
    auto step_job = [&](const JobRec &j) {  // one step as a job of its own
        const uint32_t *d_io = (const uint32_t *)(g.d_up + j.off_io);
        FlatJob f = j.ent->job;
        f.W = (uint4 *)(g.d_arena + j.off_w);
        f.T = eval && !j.rows_in_arena && !j.d_block ? (uint4 *)(g.d_up + j.off_rows) : (uint4 *)(g.d_arena + j.off_t);
        f.R = d_R;
        f.Rout = nullptr;
        f.rk = d_rk;
        f.store = store.d;
        f.in_idx = d_io;
        f.out_slots = j.ent->circ->d_out_slots;
        f.out_idx = eval ? d_io + j.nin : d_io + j.nin + j.nout;
        f.nout = j.nout;
        return f;
    };
    auto step_fin = [&](const JobRec &j, uint32_t k, const uint4 *T, const uint32_t *row_of_gate) {
        FinJob q{};
        if (!eval) {
            const uint32_t *d_io = (const uint32_t *)(g.d_up + j.off_io);
            q.a.gw = j.ent->circ->d_gwires;
            q.a.ops = j.ent->circ->d_ops;
            q.a.row_of_gate = row_of_gate;
            q.a.in = d_io;
            q.a.out = d_io + j.nin;
            q.a.ngates = j.ngates;
            q.a.first_tmp = j.first_tmp;
            q.a.first_out = j.first_out;
            q.bytes = g.d_down + sizes_bytes + j.off_bytes;
            q.size_out = (uint32_t *)g.d_down + k;
        }
        q.T = T;
        fin[k] = q;
    };
    size_t maps = off_maps, arena = arena_chain;
    uint32_t ncp = 0;
    struct DfRec {  // a record's wire maps in the upload region (host side), for its dataflow block
        uint32_t rec, nin, nout;
        size_t off_in, off_out;
    };
    std::vector<DfRec> dfrecs;
    if (df) dfrecs.reserve(nrec);
    std::vector<uint32_t> out_base;  // (chains: where each step's outputs start in the merged list; indexed by step)
    if (!g.kills.empty()) out_base.assign(n, 0xffffffffu);
    for (uint32_t u = 0; u < nwg; u++) {
        const WgRec &w = g.wgs[u];
        first[u] = rec;
        if (w.n < 2 || !plans[u]) {  // one step — or a chain without a merged plan: its steps' jobs, one after the other
            uint32_t r = 0;
            for (int32_t k = (int32_t)w.head; k >= 0; k = g.jobs[(size_t)k].next, r++) {
                const JobRec &j = g.jobs[(size_t)k];
                FlatJob f = step_job(j);
                f.pad_ = r == 0 ? w.n - 1 : 0;
                f.pad2_ = r;
                if (df) dfrecs.push_back(DfRec{rec, j.nin, j.nout, j.off_io, j.off_io + ((size_t)j.nin + j.nout) * sizeof(uint32_t)});
                fj[rec++] = f;
                if (eval && j.d_block && j.nrows) rgp[ngt++] = RowGather{j.d_block, j.d_row_off, f.T, j.nrows, 0};
                step_fin(j, (uint32_t)k, f.T, j.ent->circ->d_row_of_gate);
            }
            continue;
        }
        const FusedPlan &fp = *plans[u];
        uint32_t *h_in = (uint32_t *)(g.h_up + maps), *h_out = h_in + fp.n_ext;
        const uint32_t *d_in = (const uint32_t *)(g.d_up + maps);
        maps += up16(((size_t)fp.n_ext + fp.n_out) * sizeof(uint32_t));
        FlatJob f = fp.job;
        f.W = (uint4 *)(g.d_arena + arena);
        arena += up256(f.w_tile * 16);
        f.T = (uint4 *)(g.d_arena + arena);
        arena += up256(f.t_tile * 16);
        f.R = d_R;
        f.Rout = nullptr;
        f.rk = d_rk;
        f.store = store.d;
        f.in_idx = d_in;
        f.out_slots = fp.circ->d_out_slots;
        f.out_idx = d_in + fp.n_ext;
        f.nout = fp.n_out;
        f.pad_ = f.pad2_ = 0;
        if (df) dfrecs.push_back(DfRec{rec, fp.n_ext, fp.n_out, (size_t)((uint8_t *)h_in - g.h_up), (size_t)((uint8_t *)h_out - g.h_up)});
        fj[rec++] = f;
        uint32_t ni = 0, no = 0, m = 0;
        for (int32_t k = (int32_t)w.head; k >= 0; k = g.jobs[(size_t)k].next, m++) {
            const JobRec &j = g.jobs[(size_t)k];
            const uint32_t *io = (const uint32_t *)(g.h_up + j.off_io);
            const uint32_t *wiring = m ? g.wiring.data() + j.off_wiring : nullptr;
            for (uint32_t i = 0; i < j.nin; i++)
                if (!wiring || wiring[i] == kFuseNone) h_in[ni++] = io[i];
            const uint32_t *skip = eval ? io + j.nin : io + j.nin + j.nout;
            if (j.nout) std::memcpy(h_out + no, skip, (size_t)j.nout * sizeof(uint32_t));
            if (!out_base.empty()) out_base[(size_t)k] = (uint32_t)(h_out + no - (uint32_t *)g.h_up);
            no += j.nout;
            step_fin(j, (uint32_t)k, f.T, fp.circ->d_row_of_gate + fp.gate_base[m]);
            if (eval && j.nrows) {
                if (j.d_block) rgp[ngt++] = RowGather{j.d_block, j.d_row_off, f.T + fp.row_base[m], j.nrows, 0};
                else rcp[ncp++] = RowCopy{(const uint4 *)(g.d_up + j.off_rows), f.T + fp.row_base[m], j.nrows, 0};
            }
        }
    }
    for (const auto &kl : g.kills)  // an output a later step of the same chain writes again
        if (kl.first < n && out_base[kl.first] != 0xffffffffu) ((uint32_t *)g.h_up)[out_base[kl.first] + kl.second] = 0xffffffffu;
    if (df) {
        // The dataflow blocks, in record order — the order in which the host's counts advance IS the order of the launches, which
        // is program order for every two units that share a wire.  A record's reads first (it gathers before it stores); a killed
        // output (0xffffffff) is no store.
        size_t o = off_df;
        uint32_t *d_err = gc_ctx_err_word(ctx);
        for (const DfRec &r : dfrecs) {
            DfBlock *b = (DfBlock *)(g.h_up + o);
            uint32_t *vin = (uint32_t *)(b + 1), *req = vin + r.nin;
            const uint32_t *in_idx = (const uint32_t *)(g.h_up + r.off_in), *out_idx = (const uint32_t *)(g.h_up + r.off_out);
            *b = DfBlock{df->d_ver, df->d_rd, d_err, r.nin, r.nout, {0, 0}};
            for (uint32_t i = 0; i < r.nin; i += 4) __builtin_prefetch(&df->vr[in_idx[i]]);  // (consecutive ids share lines)
            for (uint32_t k = 0; k < r.nout; k += 4)
                if (out_idx[k] != 0xffffffffu) __builtin_prefetch(&df->vr[out_idx[k]]);
            for (uint32_t i = 0; i < r.nin; i++) {
                Dataflow::VR &c = df->vr[in_idx[i]];
                vin[i] = c.ver;
                c.rd++;
            }
            for (uint32_t k = 0; k < r.nout; k++) {
                const uint32_t w = out_idx[k];
                if (w == 0xffffffffu) {
                    req[3 * k] = req[3 * k + 1] = req[3 * k + 2] = 0;
                    continue;
                }
                Dataflow::VR &c = df->vr[w];
                req[3 * k] = c.ver, req[3 * k + 1] = c.rd, req[3 * k + 2] = ++c.ver;
            }
            fj[r.rec].prof = (uint64_t *)(g.d_up + o);
            o += up16(sizeof(DfBlock) + ((size_t)r.nin + 3 * (size_t)r.nout) * sizeof(uint32_t));
        }
        if (o > total_up) return fail("launch_group (dataflow blocks)", hipErrorInvalidValue);
    }
    if (g.has_waits) {
        uint32_t *sy = (uint32_t *)(g.h_up + off_sync);
        std::memset(sy, 0, (kSyncHead + (size_t)nwg) * sizeof(uint32_t));
        sy[1] = nwg;
        uint32_t *d_err = gc_ctx_err_word(ctx);
        std::memcpy(sy + 2, &d_err, sizeof d_err);
        static_assert(sizeof(uint32_t *) == 2 * sizeof(uint32_t) && kSyncHead == 4, "layout of the sync block's head");
        uint32_t *off = sy + kSyncHead + nwg, *list = off + nwg + 1, nl = 0;
        for (uint32_t u = 0; u < nwg; u++) {
            const WgRec &w = g.wgs[u];
            off[u] = nl;
            if (w.dep_n == kDepsAllEarlier) {
                for (uint32_t v = 0; v < u; v++) list[nl++] = v;
            } else {
                for (uint32_t k = 0; k < w.dep_n; k++) {
                    const uint32_t v = g.unit_deps[w.dep_off + k];
                    if (v >= u) return fail("launch_group (a unit that waits for a later one)", hipErrorInvalidValue);
                    list[nl++] = v;
                }
            }
        }
        off[nwg] = nl;
    }
    GroupTimeline::G *tl = nullptr;
    if (GroupTimeline::enabled() && !eval && !on_lane) {
        GroupTimeline &t = group_timeline();
        if (!t.ref) {
            (void)hipEventCreate(&t.ref);
            (void)hipEventRecord(t.ref, st);
            (void)hipEventSynchronize(t.ref);
            t.t0 = std::chrono::steady_clock::now();
        }
        t.groups.emplace_back();
        tl = &t.groups.back();
        tl->launch_no = g.launch_no, tl->steps = n, tl->h_l0 = t.now();
        (void)hipEventCreate(&tl->k0);
        (void)hipEventCreate(&tl->k1);
        (void)hipEventCreate(&tl->d);
        (void)hipEventRecord(tl->k0, st);
    }
    if (pool) {
        // Out-of-order issue: the records go up and the units are PUBLISHED on the upload stream, in launch order — behind what the
        // ctx stream did to the store outside group launches, because a workgroup of ANY launch may claim them at once —, then this
        // launch's workgroups join the pool on their own stream; the group's serialiser waits for the group's count of done units.
        if (g.pool_id == 0xffffffffu) {
            if (df->next_group >= kPoolGroups) return fail("launch_group (pool groups)", hipErrorOutOfMemory);
            g.pool_id = df->next_group++;
        }
        PoolEntry *pe = (PoolEntry *)(g.h_up + off_pool);
        for (uint32_t u = 0; u < nwg; u++) pe[u] = PoolEntry{(const FlatJob *)(g.d_up + off_fj) + first[u], fj[first[u]].pad_, g.pool_id};
        hipStream_t us = df->up_stream;
        df->pool_rounds = rounds;
        if (store.up_ev) e = hipStreamWaitEvent(us, store.up_ev, 0);
        if (e == hipSuccess && df->ctx_ev_set) e = hipStreamWaitEvent(us, df->ctx_ev, 0);
        if (e == hipSuccess) e = df->pool_start(kFlatLdsBytes);  // (persistent workgroups: behind a big step's pass, which wants its XCD)
        if (e == hipSuccess) e = hipMemcpyAsync(g.d_up, g.h_up, total_up, hipMemcpyHostToDevice, us);
        if (e == hipSuccess && df->published == 0) {  // (once: where a wait that runs out is reported)
            uint32_t *d_err = gc_ctx_err_word(ctx);
            e = hipMemcpy(&df->d_pool->host_err, &d_err, sizeof d_err, hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) {
            launch_pool_publish(df->d_pool, (const PoolEntry *)(g.d_up + off_pool), df->published, nwg, us);
            df->published += nwg;
            e = hipGetLastError();
        }
        if (e == hipSuccess && std::getenv("GC_DF_DEBUG")) {  // developer aid: what the ring holds against what was meant
            (void)hipStreamSynchronize(us);
            uint32_t head[4];
            (void)hipMemcpy(head, df->d_pool, sizeof head, hipMemcpyDeviceToHost);
            std::fprintf(stderr, "[df pool] published %u (+%u) head %u tail %u done_total %u; d_up %p off_fj %zu off_pool %zu total_up %zu\n",
                         df->published, nwg, head[0], head[1], head[2], (void *)g.d_up, off_fj, off_pool, total_up);
            {
                const FlatJob &f0 = fj[first[0]];
                std::fprintf(stderr, "[df pool]   record 0: prog %p units %p hgslot %p ogslot %p in_lds %p W %p R %p T %p rk %p te0 %p prof %p rnd %p Rout %p store %p in_idx %p out_slots %p out_idx %p nunits %u ninputs %u nout %u pad %u/%u\n",
                             (const void *)f0.prog, (const void *)f0.units, (const void *)f0.hgslot, (const void *)f0.ogslot, (const void *)f0.in_lds,
                             (void *)f0.W, (const void *)f0.R, (void *)f0.T, (const void *)f0.rk, (const void *)f0.te0, (void *)f0.prof, (const void *)f0.rnd,
                             (void *)f0.Rout, (void *)f0.store, (const void *)f0.in_idx, (const void *)f0.out_slots, (const void *)f0.out_idx, f0.nunits,
                             f0.ninputs, f0.nout, f0.pad_, f0.pad2_);
                if (std::getenv("GC_DF_DEBUG")[0] == '2') {
                    const uint32_t one = 1;
                    (void)hipMemcpy(&df->d_pool->pad_, &one, 4, hipMemcpyHostToDevice);
                }
            }
            for (uint32_t u = 0; u < std::min(nwg, 4u); u++) {
                PoolEntry back{};
                (void)hipMemcpy(&back, &df->d_pool->ring[(df->published - nwg + u) & (kPoolRing - 1u)], sizeof back, hipMemcpyDeviceToHost);
                std::fprintf(stderr, "[df pool]   unit %u: ring rec %p more %u group %u | meant rec %p more %u group %u\n", u, (const void *)back.rec,
                             back.more, back.group, (const void *)pe[u].rec, pe[u].more, pe[u].group);
            }
        }
        if (e == hipSuccess && !g.dep) e = hipEventCreateWithFlags(&g.dep, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(g.dep, us);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, g.dep, 0);
        // (persistent workgroups serve every publication: no workgroups of this launch's own)
        if (e == hipSuccess && !df->persist) e = launch_fused_flat_pool(rounds, df->d_pool, nwg, kFlatLdsBytes, st);
        if (e == hipSuccess && copy_stream != st) e = hipStreamWaitEvent(copy_stream, g.dep, 0);  // (the wait below reads the counter of THIS launch's units)
        g.pool_done += nwg;
        if (e == hipSuccess) {
            launch_pool_wait(df->d_pool->done + g.pool_id, g.pool_done, gc_ctx_err_word(ctx), copy_stream);
            e = hipGetLastError();
        }
        g.kernel_ev = g.kdone;
        if (e == hipSuccess) e = hipEventRecord(g.kdone, copy_stream);
        df->n_launches++;
        if (e == hipSuccess && on_lane) {
            deep.inflight[(size_t)g.lane].push_back(DeepLanes::InFlight{g.deep_id, g.kernel_ev});
            deep.n_inflight++;
            deep.n_steps++;
        }
    } else {
    e = hipMemcpyAsync(g.d_up, g.h_up, total_up, hipMemcpyHostToDevice, st);  // pinned source: a true asynchronous copy
    if (e == hipSuccess && g.rows_ev) e = hipStreamWaitEvent(st, g.rows_ev, 0);
    if (e == hipSuccess && ncp) {
        hipLaunchKernelGGL(k_rows_copy, dim3(ncp), dim3(256), 0, st, (const RowCopy *)(g.d_up + off_rc));
        e = hipGetLastError();
    }
    if (e == hipSuccess && ngt) {
        hipLaunchKernelGGL(k_rows_gather, dim3(ngt), dim3(256), 0, st, (const RowGather *)(g.d_up + off_rg));
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = launch_fused_flat_jobs(eval, rounds, has_or, (const FlatJob *)(g.d_up + off_fj),
                                   nrec != nwg || g.has_waits ? (const uint32_t *)(g.d_up + off_first) : nullptr, nwg, lds, st,
                                   g.has_waits ? (uint32_t *)(g.d_up + off_sync) : nullptr);
    // "the group's kernel has run": kdone for the garbler (the serialiser and the bytes' way back follow on the copy stream),
    // done itself for the evaluator (nothing follows)
    g.kernel_ev = eval ? g.done : g.kdone;
    if (tl) (void)hipEventRecord(tl->k1, st);
    if (e == hipSuccess) e = hipEventRecord(g.kernel_ev, st);
    if (e == hipSuccess && df) {
        if (!on_lane) {
            e = hipEventRecord(df->tail[df_k], st);
            df->used[df_k] = 1;
        }
        df->inflight.push_back(Dataflow::InFlight{&g, g.launch_no, nwg});
        df->account(nwg, true);
        df->n_launches++;
    }
    if (e == hipSuccess && on_lane) {
        deep.inflight[(size_t)g.lane].push_back(DeepLanes::InFlight{g.deep_id, g.kernel_ev});
        deep.n_inflight++;
        deep.n_steps++;
    }
    }  // (!pool)
    if (e == hipSuccess && !eval) {
        // serialiser and bytes on the copy stream: the next group's garbling need not wait for either
        if (copy_stream != st && !pool) e = hipStreamWaitEvent(copy_stream, g.kdone, 0);
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
                ser_sizes_scan(q.a, boff, nblocks, q.size_out, copy_stream);
                ser_write(q.a, boff, nblocks, q.T, dense, q.bytes, copy_stream);
                e = hipGetLastError();
            }
        } else if (e == hipSuccess) {
            uint32_t max_gates = 0;
            for (const JobRec &j : g.jobs) max_gates = std::max(max_gates, j.ngates);
            ser_group((const FinJob *)(g.d_up + off_fin), n, max_gates, copy_stream);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(g.h_down, g.d_down, sizes_bytes + g.down_used, hipMemcpyDeviceToHost, copy_stream);
        if (e == hipSuccess) e = hipEventRecord(g.done, copy_stream);
        if (tl) (void)hipEventRecord(tl->d, copy_stream);
    }
    if (tl) tl->h_l1 = group_timeline().now();
    if (e != hipSuccess) return fail("launch_group", e);
    return GC_OK;
}

}  // namespace gcs
