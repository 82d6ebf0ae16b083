// stream_eval.cpp — C ABI of the streaming evaluator (circuit.StreamEval): circuit/stream_evaluator.go:29-96
// (the wire store) and :226-432 (one OpCircuit block: parse, evaluate, Set the global outputs).  Blocks are grouped, laned and
// launched like the garbler's steps (stream_garble.cpp's head, stream_group.cpp); a block seen before is recognised by its byte
// skeleton (stream_skel.h).
#include "stream_eval_internal.h"

using namespace gcs;

namespace {

int eval_launch_oldest(gc_stream_eval *e) {
    if (e->win.open.empty()) return GC_OK;
    const uint32_t seq = e->win.first_seq, slot = e->win.pop();
    Slot &g = *e->slots[slot];
    e->n_groups++;
    e->n_group_blocks += g.jobs.size();
    e->prof.lap(StageProf::kOther);
    const int rc = launch_group(e->ctx, g, true, e->store, e->d_rk, nullptr, e->rounds, nullptr, e->deep, &e->fuse);
    e->prof.lap(StageProf::kLaunch);
    if (!g.chunk_refs.empty()) evdev_launched(e, g, slot);
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
    e->prof.print("evaluator", e->n_blocks_total);
    if (e->ctx) {
        (void)hipSetDevice(e->ctx->device);
        (void)hipStreamSynchronize(e->ctx->stream);
    }
    e->deep.release();
    evdev_free(e);
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

int gc_stream_eval_dev_stats(const gc_stream_eval *e, uint64_t *blocks, uint64_t *fallbacks) {
    if (!e) return GC_E_ARG;
    evdev_stats(e, blocks, fallbacks);
    return GC_OK;
}

int gc_stream_eval_fuse_stats(const gc_stream_eval *e, uint64_t *fused_units, uint64_t *fused_blocks, uint64_t *plans_built,
                              uint64_t *unfit) {
    if (!e) return GC_E_ARG;
    if (fused_units) *fused_units = e->fuse.units;
    if (fused_blocks) *fused_blocks = e->fuse.steps;
    if (plans_built) *plans_built = e->fuse.built;
    if (unfit) *unfit = e->fuse.unfit;
    return GC_OK;
}

int gc_stream_eval_wait_stats(const gc_stream_eval *e, uint64_t *waiting_units) {
    if (!e) return GC_E_ARG;
    if (waiting_units) *waiting_units = e->fuse.waiting;
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
    const int rcc = gc_ctx_coop_check(e->ctx);  // (the stream was waited for: a cooperative pass that lost a workgroup is noted here)
    return rc != GC_OK ? rc : rcc;
} catch (...) {
    return gc::on_exception();
}

}  // extern "C"

namespace gcs {

uint32_t stream_max_wires() {
    static const uint32_t v = [] {
        const char *e = std::getenv("GC_STREAM_MAX_WIRES");
        const unsigned long long n = e ? std::strtoull(e, nullptr, 0) : 0;
        return n ? (uint32_t)std::min<unsigned long long>(n, 0xffffffffull) : (1u << 28);
    }();
    return v;
}

bool eval_adopt_ids(gc_stream_eval *e, const EvalSkel &sk, const uint32_t *ids, uint32_t nwires) {
    if (++e->gen == 0) {
        std::fill(e->last_t.begin(), e->last_t.end(), 0);
        std::fill(e->last_w.begin(), e->last_w.end(), 0);
        e->gen = 1;
    }
    const uint64_t stamp = (uint64_t)e->gen << 32;
    const uint32_t nf = (uint32_t)sk.gf_off.size();
    for (uint32_t f = 0; f < nf; f++) {
        const uint32_t id = ids[f];
        if (id >= nwires) return false;
        if (id >= e->last_w.size()) e->last_w.resize((size_t)id + 1 + e->last_w.size() / 2, 0);
        uint32_t first = f;
        if ((e->last_w[id] >> 32) == e->gen) first = (uint32_t)e->last_w[id];
        else e->last_w[id] = stamp | f;
        if (sk.gf_canon[f] != first) return false;
    }
    const uint32_t nin = sk.nin, nout = sk.nout;
    e->io_host.resize((size_t)nin + nout + 1);
    for (uint32_t k = 0; k < nin; k++) e->io_host[k] = ids[sk.in_gf[k]];
    e->wr_ids.resize(nout);
    for (uint32_t k = 0; k < nout; k++) {
        e->wr_ids[k] = ids[sk.out_gf[k]];
        e->io_host[nin + k] = sk.out_live[k] ? e->wr_ids[k] : 0xffffffffu;
    }
    return true;
}

// One OpCircuit block (gc_stream_eval_circuit; block after block in gc_stream_eval_blocks)
// the row buffer of the block being taken in: a small block (candidate for a step group) parses its rows into plain host
// scratch — they are copied into the group's pinned upload region when the block is queued —, a big block uses the next entry
// of the pinned ring
int eval_rows_buffer(gc_stream_eval *e, uint32_t ngates, bool *small_block_out, uint32_t *sb_out, gc_label **slab_out) {
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
    *small_block_out = small_block, *sb_out = sb, *slab_out = slab;
    return GC_OK;
}

// The block joins a group / takes a lane / runs as a pass of its own (see the head of stream_garble.cpp)
int eval_schedule(gc_stream_eval *e, const BlockIn &in, size_t *consumed) {
    CircEntry *ent = in.ent;
    const uint32_t ngates = in.ngates, nin = in.nin, nout = in.nout, sb = in.sb;
    const size_t nrows = in.nrows, pos = in.pos;
    const bool small_block = in.small_block;
    gc_label *slab = in.slab;
    const EvalSkel *rows_from = in.rows_from;
    const uint8_t *buf = in.buf;
    std::vector<uint32_t> &wr_ids = e->wr_ids;
    StreamTrace tr;
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
    e->n_blocks_total++;
    e->prof.lap(StageProf::kOther);
    if (is_deep || (small_block && entry_is_small(ent))) {
        if (!e->store.dirty.empty()) {  // host-set labels go up before the block's outputs are marked device-owned (and before
            std::lock_guard<std::mutex> lk(ctx->mu);  // the block is put anywhere: a failure leaves nothing half-queued)
            int rcs = e->store.flush(ctx);
            if (rcs != GC_OK) return rcs;
        }
        e->win.ensure(e->store.host.size());
        if (is_deep || e->deep.n_inflight) e->deep.ensure(e->store.host.size());
        const size_t wbytes = up256((size_t)ent->job.w_tile * 16);
        // chain fusion, as in the garbler (stream_garble.cpp: stream_begin): a short block whose conflicts with the latest group
        // it has any with all sit in ONE launch unit is appended to that unit
        uint32_t unit = kFuseNone;
        uint32_t gi = fuse_enabled() ? e->win.place_fuse(e->io_host.data(), nin, wr_ids.data(), nout, &unit)
                                     : e->win.place(e->io_host.data(), nin, wr_ids.data(), nout);
        const bool may_fuse = fuse_enabled() && !is_deep && small_block && ent->uid != 0;
        // conflicts with several units of that group (as in the garbler): appended to the latest if that one starts behind all
        // the others, else a unit that waits
        uint32_t dep_units[kUnitDeps], ndeps = 0;
        const bool can_wait = fuse_enabled() && e->use_deps && !is_deep && small_block && gi > 0;
        if (can_wait && unit == kFuseMulti) {
            ndeps = e->win.conflict_units(e->io_host.data(), nin, wr_ids.data(), nout, e->win.first_seq + gi - 1, dep_units, kUnitDeps);
            unit = wg_covering(*e->slots[e->win.open[gi - 1]], dep_units, ndeps);
        } else if (can_wait) {
            dep_units[0] = unit, ndeps = 1;
        }
        bool fuse = false;
        uint64_t shape = 0;
        uint32_t n_ext = 0;
        if (may_fuse && gi > 0 && unit < kFuseMulti && ngates <= kFuseTailGates) {
            const Slot &fg = *e->slots[e->win.open[gi - 1]];
            const WgRec &w = fg.wgs[unit];
            fuse = w.open && w.n < kFuseMembers && w.gates + ngates <= kFuseGates && w.slots + ent->job.zslot + 1 <= kFuseSlots &&
                   w.inputs + nin <= kFuseInputs && fg.jobs.size() < kGroupSteps &&
                   fg.arena_used + fg.up_used + 2 * wbytes + 2 * nrows * 16 <= kGroupBytes;
            if (fuse) {  // input sources, and what chains of this shape say about their depth (as in the garbler)
                const uint32_t seq = e->win.first_seq + gi - 1;
                e->wiring_scratch.resize(nin);
                for (uint32_t i = 0; i < nin; i++) {
                    const GroupWindow::WireRec &r = e->win.rec[e->io_host[i]];
                    if (r.wr == seq && r.wrj == unit) e->wiring_scratch[i] = (fg.jobs[r.wrm >> 20].member << 24) | (r.wrm & 0xfffffu);
                    else e->wiring_scratch[i] = kFuseNone, n_ext++;
                }
                shape = fuse_shape(w.shape, ent, e->wiring_scratch.data());
                const uint32_t hint = fuse_depth_hint(ctx, shape);
                fuse = hint ? hint <= fuse_depth_cap() : w.depth_sum + ent->circ->plan.p.n_hash_phases <= 2 * fuse_depth_cap();
            }
        }
        uint32_t slot_idx = 0;
        auto full = [&](const Slot &g) {
            return g.wgs.size() >= kGroupJobs || g.jobs.size() >= kGroupSteps ||
                   g.arena_used + g.up_used + 2 * wbytes + 2 * nrows * 16 > kGroupBytes;
        };
        // not fused: the block still joins the group it conflicts with, as a unit that waits on the device for the units it
        // conflicts with there (as in the garbler)
        bool waits = false;
        if (!fuse && can_wait && !full(*e->slots[e->win.open[gi - 1]])) {
            waits = true;
            gi--;
        }
        e->prof.lap(StageProf::kPlace);
        if (fuse || waits) {
            if (fuse) gi--;
            slot_idx = e->win.open[gi];
        } else if (is_deep) {
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
        // the rows of a block of many gates were parsed into the pinned ring: they go up from there (launch_group); the rows of
        // a block the DEVICE matched are in device memory already: the launch sequence gathers them into the job's table array
        const bool ext_rows = is_deep && !small_block;
        const bool dev_rows = in.d_block != nullptr && !ext_rows;
        const size_t io_bytes = up16(((size_t)nin + nout) * sizeof(uint32_t)), row_bytes = ext_rows || dev_rows ? 0 : up16(nrows * sizeof(gc_label));
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
        if (!ext_rows && !dev_rows && nrows) {
            if (rows_from) rows_from->copy_rows(buf, (gc_label *)(g.h_up + g.up_used));
            else std::memcpy(g.h_up + g.up_used, slab, nrows * sizeof(gc_label));
        }
        g.up_used += row_bytes;
        j.off_w = g.arena_used;
        g.arena_used += wbytes;
        if (dev_rows) {
            j.d_block = in.d_block, j.d_row_off = in.d_row_off, j.chunk = in.chunk;
            j.off_t = g.arena_used;
            g.arena_used += up256(nrows * sizeof(gc_label));
            g.chunk_refs.push_back(in.chunk);
            evdev_ref(e, in.chunk);
        }
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
        j.nrows = (uint32_t)nrows;
        const uint32_t step_idx = (uint32_t)g.jobs.size();
        if (fuse) {  // the wiring goes with the block; outputs of earlier blocks of the unit that this one overwrites
            const uint32_t seq = e->win.first_seq + gi;
            j.off_wiring = g.wiring.size();
            g.wiring.insert(g.wiring.end(), e->wiring_scratch.begin(), e->wiring_scratch.begin() + nin);
            for (uint32_t k = 0; k < nout; k++) {
                if (e->io_host[nin + k] == 0xffffffffu) continue;  // (superseded inside the block itself: not stored anyway)
                const GroupWindow::WireRec &r = e->win.rec[wr_ids[k]];
                if (r.wr == seq && r.wrj == unit) g.kills.emplace_back(r.wrm >> 20, r.wrm & 0xfffffu);
            }
            wg_append(g, unit, &j, ent, n_ext, shape);
            e->fuse.appended++;
        } else {
            unit = wg_new(g, &j, ent, may_fuse);
            if (waits) {
                wg_wait(g, unit, dep_units, ndeps);
                e->fuse.waiting++;
            }
        }
        g.jobs.push_back(j);
        e->prof.lap(StageProf::kQueue);
        if (!is_deep) {
            e->win.mark(gi, e->io_host.data(), nin, wr_ids.data(), nout, unit, step_idx);
            if (e->deep.n_inflight) g.deps.merge(e->deep.conflicts(e->io_host.data(), nin, wr_ids.data(), nout));
        }
        if (is_deep) {  // launched at once, on its lane
            int rcl = launch_group(ctx, g, true, e->store, e->d_rk, nullptr, e->rounds, nullptr, e->deep, &e->fuse);
            if (!g.chunk_refs.empty()) evdev_launched(e, g, slot_idx);
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
        e->prof.lap(StageProf::kMark);
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
    bool small_block = false;
    uint32_t sb = 0;
    gc_label *slab = nullptr;
    {
        int rcb = eval_rows_buffer(e, ngates, &small_block, &sb, &slab);
        if (rcb != GC_OK) return rcb;
    }
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
                            evdev_drop(e, v[i].get());
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
        ent->eval_form = true;  // (chain fusion: `gates` holds the block as parsed; its ctx-wide identity)
        ent->uid = fuse_register(e->ctx, true, h, ent->gates, cw, nin, nout);
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
            evdev_drop(e, v.back().get());
            e->skel_bytes -= std::min(e->skel_bytes, v.back()->held());
            v.pop_back();
        }
        std::unique_ptr<EvalSkel> sk(new EvalSkel);
        sk->nbytes = pos;
        sk->key = ((uint64_t)ngates << 32) | ntmp;
        sk->nrows = (uint32_t)nrows, sk->nin = nin, sk->nout = nout;
        sk->ent = ent;
        sk->bytes.assign(buf, buf + pos);
        sk->build(rec.chunks);
        sk->gf_off = rec.gf_off;
        sk->in_gf = rec.in_gf, sk->out_gf = rec.out_gf, sk->out_live = rec.out_live;
        if (canon_of(sk->gf_off, &gf_ids, nullptr, &sk->gf_canon)) {
            sk->shape_hash();
            e->skel_bytes += sk->held();
            evdev_add(e, sk.get());
            v.insert(v.begin(), std::move(sk));
        }
    }
    }  // parsed
    return eval_schedule(e, BlockIn{ent, ngates, nin, nout, nrows, pos, small_block, sb, slab, rows_from, buf}, consumed);
}

}  // namespace gcs

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
    // a read buffer of many blocks: the device recognises them (stream_eval_dev.cpp) — what it leaves (a cut-off last block,
    // a block nobody has seen yet that is not all here, another operation) goes through the loop below
    // (the device's turn again behind every block the host had to take: it only stops where it cannot place a block)
    auto device_turn = [&]() {
        size_t used = 0;
        uint32_t nb = 0;
        rc = evdev_blocks(e, buf + pos, len - pos, &used, &nb);
        pos += used, n += nb;
    };
    device_turn();
    while (rc == GC_OK && len - pos >= 20 && be32(buf + pos) == 1 /* OpCircuit, stream_evaluator.go:22-26 */) {
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
        device_turn();
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
