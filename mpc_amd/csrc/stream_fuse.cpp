// stream_fuse.cpp — chain fusion of the streaming engine ("super-steps").
//
// A compiled program is a long sequence of small circuits, one per SSA instruction (compiler/ssa/streamer.go:412-524), and
// its critical path is chains of them: h = f0*g0 + f1*g9 + ... compiles to mul, mul, add, mul, add, ... where every add waits
// for the add before it.  Run step by step — one launch unit per step, a group per link of the chain — the ten-link add chain
// of an Ed25519 field multiplication costs ten full ripple-carry adders (10 x 63 dependent hash phases, each a lone AES on
// one CU) while 250 CUs idle.  Scheduled at GATE granularity across the step boundaries the same chain costs 63 + 9 phases:
// bit i of link k + 1 only needs bit i of link k.
//
// So a queued step whose dependencies inside the window all sit in ONE launch unit of the latest group it conflicts with is
// APPENDED to that unit (stream_garble.cpp: stream_begin; stream_eval.cpp: eval_block), and when the group is launched a unit
// of several steps runs as ONE planned job: the steps' gate lists concatenated with the wires re-named (an input that an
// earlier step of the chain wrote IS that step's output wire, everything else is read from the wire store), planned by the
// same planner as any circuit (plan.cpp) with the two things the reference derives from its serial order frozen per gate as
// always — the hash tweak, which Streaming.Garble starts over at every circuit (circuit/stream_garble.go:174, the evaluator
// :270), and the table rows, which the garbler's serialiser still finds step by step (row_base) so that the bytes leave per
// step, in program order, exactly as circuit/stream_garble.go:385-449 writes them.  Results are bit-identical because the
// labels of a wire do not depend on WHEN its gate runs.
//
// Merged plans are cached per ctx, keyed by the chain's circuits (a ctx-wide registry gives every circuit content an id; a
// hit is only taken on an exact key match) and its wiring: a compiled program repeats its chains thousands of times.  A chain
// whose merged plan does not fit one workgroup's LDS is remembered as such and its steps run one launch after the other.
#include <unordered_set>

#include "stream_internal.h"

using namespace gcs;

struct gcs_fuse_todo {  // one chain to plan, self-contained (the steps' circuits belong to a stream's cache and may go first)
    uint64_t h;
    std::vector<uint32_t> key;
    std::vector<std::vector<gc_gate>> gates;
    std::vector<std::vector<uint32_t>> wiring;
    std::vector<uint32_t> uid, nwires, nin, nout;
};

struct gcs_fuse_cache {
    using Todo = gcs_fuse_todo;
    struct Reg {
        uint32_t uid;
        bool eval_form;
        uint32_t nwires, nin, nout;
        std::vector<CircKey> gates;
    };
    std::unordered_multimap<uint64_t, Reg> registry;
    size_t reg_gates = 0;
    uint32_t next_uid = 1;
    std::unordered_multimap<uint64_t, std::unique_ptr<FusedPlan>> plans;
    size_t plan_gates = 0;
    std::unordered_map<uint64_t, uint32_t> depth;  // shape of a chain's prefix -> hash phases of its merged plan (fuse_depth_hint)
    std::unordered_map<uint64_t, uint32_t> seen;   // key hash of a chain without a plan yet -> times it was met
    // the planner: chains to plan, the ones in its hands (by key hash), its thread
    std::deque<Todo> todo;
    size_t todo_gates = 0;
    std::unordered_set<uint64_t> pending;
    std::thread planner;
    std::condition_variable cv;
    bool stop = false;
    uint64_t built = 0;
};

void gcs_fuse_cache_free(gc_ctx *ctx) {  // (gc_ctx_destroy: nobody else holds the ctx any more)
    gcs_fuse_cache *f = ctx->fuse;
    if (!f) return;
    if (f->planner.joinable()) {
        {
            std::lock_guard<std::mutex> lk(ctx->fuse_mu);
            f->stop = true;
        }
        f->cv.notify_all();
        f->planner.join();
    }
    ctx->fuse = nullptr;
    for (auto &kv : f->plans)
        if (kv.second->circ) gc_circ_free(kv.second->circ);
    delete f;
}

namespace gcs {

constexpr uint32_t kFuseSightings = 3;              // a chain is planned when it is met this often
constexpr size_t kRegistryGates = (size_t)4 << 20;  // gate records the registry may hold per ctx (16 B each)
constexpr size_t kPlanGates = (size_t)8 << 20;      // gates of all merged plans of a ctx (beyond it new chains run step by step)

bool fuse_enabled() {
    static const bool on = [] {
        const char *v = std::getenv("GC_STREAM_NO_FUSE");
        return !(v && *v && *v != '0');
    }();
    return on;
}

static uint64_t shape_step(uint64_t shape, uint32_t uid, const uint32_t *wiring, uint32_t nin) {
    uint64_t h = (shape ^ uid) * 1099511628211ull + 0x9e3779b97f4a7c15ull;
    if (wiring)
        for (uint32_t i = 0; i < nin; i++) h = (h ^ wiring[i]) * 1099511628211ull;
    return h ^ (h >> 29);
}

static gcs_fuse_cache *cache_of(gc_ctx *ctx) {  // under ctx->fuse_mu
    if (!ctx->fuse) ctx->fuse = new (std::nothrow) gcs_fuse_cache;
    return ctx->fuse;
}

uint32_t fuse_register(gc_ctx *ctx, bool eval_form, uint64_t hash, const std::vector<CircKey> &gates, uint32_t nwires, uint32_t nin,
                       uint32_t nout) {
    if (!fuse_enabled() || gates.empty() || gates.size() > kSmallGates) return 0;
    // A caller's gate list may write one of its own INPUT wires (gc_stream_intern takes any list).  Step by step that only
    // changes the step's local wire; in a merged plan the input IS an earlier member's output wire and the write would land
    // there.  Such a circuit gets no identity: it never joins a chain (the evaluator's form names operands by the gate that
    // wrote them, so the case cannot arise there).
    if (!eval_form)
        for (const CircKey &g : gates)
            if (g.out < nin) return 0;
    std::lock_guard<std::mutex> lk(ctx->fuse_mu);
    gcs_fuse_cache *f = cache_of(ctx);
    if (!f) return 0;
    const uint64_t h = hash ^ (eval_form ? 0x9e3779b97f4a7c15ull : 0);
    auto range = f->registry.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
        const gcs_fuse_cache::Reg &r = it->second;
        if (r.eval_form == eval_form && r.nwires == nwires && r.nin == nin && r.nout == nout && r.gates.size() == gates.size() &&
            std::memcmp(r.gates.data(), gates.data(), gates.size() * sizeof(CircKey)) == 0)
            return r.uid;
    }
    if (f->reg_gates + gates.size() > kRegistryGates || f->next_uid == 0xffffffffu) return 0;
    gcs_fuse_cache::Reg r{f->next_uid++, eval_form, nwires, nin, nout, gates};
    f->reg_gates += gates.size();
    const uint32_t uid = r.uid;
    f->registry.emplace(h, std::move(r));
    return uid;
}

// One step's gates in its circuit's own wire ids: inputs [0, nin), outputs the last nout wires.  The garbler's cache holds the
// caller's gate list as it is; the evaluator's holds the block AS PARSED (stream_eval.cpp: operands named by the gate that
// wrote them, bit 31: the k-th distinct input; out: writes a tmp wire) — its circuit numbers the tmp-writing gates first and
// the gates that write a global wire last, as eval_block does for the circuit it loads.
static void member_gates(const CircEntry &e, std::vector<gc_gate> *out) {
    const uint32_t n = (uint32_t)e.gates.size();
    out->assign(n, gc_gate{});
    if (!e.eval_form) {
        for (uint32_t g = 0; g < n; g++) {
            (*out)[g].in0 = e.gates[g].in0, (*out)[g].in1 = e.gates[g].in1, (*out)[g].out = e.gates[g].out;
            (*out)[g].op = (uint8_t)e.gates[g].op;
        }
        return;
    }
    const uint32_t n_tmp = n - e.nout;
    std::vector<uint32_t> id_of(n);
    uint32_t kt = 0, kg = 0;
    for (uint32_t g = 0; g < n; g++) id_of[g] = e.gates[g].out ? e.nin + kt++ : e.nin + n_tmp + kg++;
    auto fix = [&](uint32_t v) { return (v & 0x80000000u) ? (v & 0x7fffffffu) : id_of[v]; };
    for (uint32_t g = 0; g < n; g++) {
        (*out)[g].in0 = fix(e.gates[g].in0);
        (*out)[g].in1 = e.gates[g].op == GC_INV ? (*out)[g].in0 : fix(e.gates[g].in1);
        (*out)[g].out = id_of[g];
        (*out)[g].op = (uint8_t)e.gates[g].op;
    }
}

int merge_chain(const ChainStep *steps, uint32_t n, std::vector<gc_gate> *all, std::vector<uint32_t> *gate_base, uint32_t *n_ext_out,
                uint32_t *n_tmp_out, uint32_t *n_out_out) {
    uint32_t n_ext = 0, n_tmp = 0, n_out = 0, n_gates = 0;
    std::vector<uint32_t> tmp_base(n), out_base(n);
    gate_base->clear();
    for (uint32_t k = 0; k < n; k++) {
        const ChainStep &e = steps[k];
        if ((uint64_t)e.nin + e.nout > e.nwires) return GC_E_ARG;  // (outputs that are input wires: never fused)
        tmp_base[k] = n_tmp, out_base[k] = n_out;
        gate_base->push_back(n_gates);
        n_tmp += e.nwires - e.nin - e.nout;
        n_out += e.nout;
        n_gates += e.ngates;
        for (uint32_t i = 0; i < e.nin; i++) n_ext += !e.wiring || e.wiring[i] == kFuseNone;
    }
    gate_base->push_back(n_gates);
    all->clear();
    all->reserve(n_gates);
    std::vector<uint32_t> in_map;
    uint32_t ext = 0;
    for (uint32_t k = 0; k < n; k++) {
        const ChainStep &e = steps[k];
        in_map.resize(e.nin);
        for (uint32_t i = 0; i < e.nin; i++) {
            const uint32_t src = e.wiring ? e.wiring[i] : kFuseNone;
            if (src == kFuseNone) {
                in_map[i] = ext++;
            } else {
                const uint32_t m = src >> 24, j = src & 0xffffffu;
                if (m >= k || j >= steps[m].nout) return GC_E_ARG;
                in_map[i] = n_ext + n_tmp + out_base[m] + j;
            }
        }
        const uint32_t first_out = e.nwires - e.nout;
        auto map = [&](uint32_t w) {
            if (w < e.nin) return in_map[w];
            if (w < first_out) return n_ext + tmp_base[k] + (w - e.nin);
            return n_ext + n_tmp + out_base[k] + (w - first_out);
        };
        for (uint32_t gi = 0; gi < e.ngates; gi++) {
            const gc_gate &g = e.gates[gi];
            if (g.in0 >= e.nwires || g.out >= e.nwires || (g.op != GC_INV && g.in1 >= e.nwires)) return GC_E_WIRE;
            gc_gate q{};
            q.in0 = map(g.in0);
            q.in1 = g.op == GC_INV ? q.in0 : map(g.in1);
            q.out = map(g.out);
            q.op = g.op;
            all->push_back(q);
        }
    }
    *n_ext_out = n_ext, *n_tmp_out = n_tmp, *n_out_out = n_out;
    return GC_OK;
}

// merge, plan, upload; the plan (or "no one-workgroup plan") goes into the cache under the lock.  Runs on the planner's thread
// (or on the caller's: GC_STREAM_FUSE_EAGER).
static void plan_chain(gc_ctx *ctx, gcs_fuse_cache *f, gcs_fuse_cache::Todo &t) {
    const uint32_t n = (uint32_t)t.gates.size();
    std::unique_ptr<FusedPlan> fp(new FusedPlan);
    fp->key = t.key;
    std::vector<ChainStep> steps(n);
    for (uint32_t k = 0; k < n; k++)
        steps[k] = ChainStep{t.gates[k].data(), (uint32_t)t.gates[k].size(), t.nwires[k], t.nin[k], t.nout[k], k ? t.wiring[k].data() : nullptr};
    std::vector<gc_gate> all;
    uint32_t n_ext = 0, n_tmp = 0, n_out = 0;
    std::vector<std::pair<uint64_t, uint32_t>> depths;
    if (merge_chain(steps.data(), n, &all, &fp->gate_base, &n_ext, &n_tmp, &n_out) == GC_OK) {
        const uint32_t n_gates = (uint32_t)all.size();
        fp->n_ext = n_ext, fp->n_out = n_out;
        {  // what this chain tells later ones of its shape: dependent hash phases up to the end of every step
            std::vector<uint32_t> d(n_ext + n_tmp + n_out, 0);
            uint64_t shape = 0;
            uint32_t deepest = 0;
            for (uint32_t k = 0; k < n; k++) {
                for (uint32_t gi = fp->gate_base[k]; gi < fp->gate_base[k + 1]; gi++) {
                    const gc_gate &g = all[gi];
                    const uint32_t a = std::max(d[g.in0], d[g.in1]);
                    d[g.out] = g.op == GC_XOR || g.op == GC_XNOR ? a : a + 1;
                    deepest = std::max(deepest, d[g.out]);
                }
                shape = shape_step(shape, t.uid[k], k ? t.wiring[k].data() : nullptr, t.nin[k]);
                depths.emplace_back(shape, std::max(deepest, 1u));
            }
        }
        (void)hipSetDevice(ctx->device);
        int st = GC_OK;
        gc_circ *circ = gc_circ_load_seg(ctx, all.data(), n_gates, n_ext + n_tmp + n_out, n_ext, n_out, fp->gate_base.data(), n, &st);
        if (circ && gc_circ_flat_job(circ, &fp->job, &fp->lds, &fp->has_or)) {
            fp->circ = circ;
            fp->n_steps = circ->plan.p.n_flat_steps;
            for (uint32_t k = 0; k <= n; k++) fp->row_base.push_back(circ->plan.p.row_of_gate[fp->gate_base[k]]);
        } else {  // no one-workgroup plan (or no memory for it): remembered, the chain's steps run one after the other
            if (circ) gc_circ_free(circ);
            (void)hipGetLastError();
        }
        if (std::getenv("GC_TRACE"))
            std::fprintf(stderr, "[gc trace] fused chain of %u steps, %u gates, %u inputs: %s, %u barriers, %zu B of LDS\n", n, n_gates, n_ext,
                         fp->circ ? "planned" : "NO one-workgroup plan", fp->n_steps, fp->lds);
    }
    std::lock_guard<std::mutex> lk(ctx->fuse_mu);
    for (const auto &d : depths)
        if (f->depth.size() < ((size_t)1 << 20)) f->depth[d.first] = d.second;
    if (fp->circ) f->plan_gates += fp->gate_base.back();
    f->pending.erase(t.h);
    f->plans.emplace(t.h, std::move(fp));
    f->built++;
}

static void planner_loop(gc_ctx *ctx, gcs_fuse_cache *f) {
    for (;;) {
        gcs_fuse_cache::Todo t;
        {
            std::unique_lock<std::mutex> lk(ctx->fuse_mu);
            f->cv.wait(lk, [&] { return f->stop || !f->todo.empty(); });
            if (f->stop) return;
            t = std::move(f->todo.front());
            f->todo.pop_front();
        }
        try {
            plan_chain(ctx, f, t);
        } catch (...) {  // (no memory for this one: forgotten, it may be asked for again)
            (void)gc::on_exception();
            std::lock_guard<std::mutex> lk(ctx->fuse_mu);
            f->pending.erase(t.h);
        }
    }
}

const FusedPlan *fuse_plan(gc_ctx *ctx, bool eval_form, const FuseMember *members, uint32_t n, bool *asked) {
    if (asked) *asked = false;
    if (n < 2) return nullptr;
    // the key: form, steps, then per step its circuit's id and input count, then the wiring of every step but the first
    std::vector<uint32_t> key;
    key.reserve(2 + 2 * (size_t)n + 256);
    key.push_back(eval_form ? 1u : 0u);
    key.push_back(n);
    for (uint32_t k = 0; k < n; k++) {
        if (!members[k].ent->uid) return nullptr;
        if ((uint64_t)members[k].ent->nin + members[k].ent->nout > members[k].ent->nwires) return nullptr;
        key.push_back(members[k].ent->uid);
        key.push_back(members[k].ent->nin);
    }
    for (uint32_t k = 1; k < n; k++) key.insert(key.end(), members[k].wiring, members[k].wiring + members[k].ent->nin);
    uint64_t h = 1469598103934665603ull;
    for (uint32_t w : key) h = (h ^ w) * 1099511628211ull;
    std::unique_lock<std::mutex> lk(ctx->fuse_mu);
    gcs_fuse_cache *f = cache_of(ctx);
    if (!f) return nullptr;
    {
        auto range = f->plans.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (it->second->key == key) return it->second.get();
    }
    // A chain is planned when it is met for the THIRD time — planning costs what a few hundred runs of the chain cost; a
    // compiled program repeats its chains thousands of times, a chain that never comes back (a random instruction mix, the
    // odd cut at the window's edge) is not worth a plan — and by a thread of the ctx's own: until the plan is there the chain's
    // steps run one launch after the other, and nobody waits.  (GC_STREAM_FUSE_EAGER: at first sight, on the caller's thread.)
    const bool eager = std::getenv("GC_STREAM_FUSE_EAGER") != nullptr;  // (read here, where a chain is unknown: rare)
    if (f->pending.count(h)) return nullptr;
    if (!eager) {
        if (f->seen.size() > ((size_t)1 << 20)) f->seen.clear();
        if (++f->seen[h] < kFuseSightings) return nullptr;
        f->seen.erase(h);
    }
    size_t gates = 0;
    for (uint32_t k = 0; k < n; k++) gates += members[k].ent->gates.size();
    if (f->plan_gates + f->todo_gates + gates > kPlanGates || f->todo.size() >= 256) return nullptr;
    gcs_fuse_cache::Todo t;
    t.h = h;
    t.key = std::move(key);
    t.gates.resize(n), t.wiring.resize(n);
    for (uint32_t k = 0; k < n; k++) {
        const CircEntry &e = *members[k].ent;
        member_gates(e, &t.gates[k]);
        if (k) t.wiring[k].assign(members[k].wiring, members[k].wiring + e.nin);
        t.uid.push_back(e.uid), t.nwires.push_back(e.nwires), t.nin.push_back(e.nin), t.nout.push_back(e.nout);
    }
    if (asked) *asked = true;
    f->pending.insert(h);
    if (eager) {
        lk.unlock();
        plan_chain(ctx, f, t);
        lk.lock();
        auto range = f->plans.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (it->second->key == t.key) return it->second.get();
        return nullptr;
    }
    f->todo.push_back(std::move(t));
    if (!f->planner.joinable()) {
        try {
            f->planner = std::thread(planner_loop, ctx, f);
        } catch (...) {
            f->pending.erase(h);
            f->todo.pop_back();
            return nullptr;
        }
    }
    f->cv.notify_one();
    return nullptr;
}

uint64_t fuse_shape(uint64_t shape, const CircEntry *ent, const uint32_t *wiring) { return shape_step(shape, ent->uid, wiring, ent->nin); }

uint32_t fuse_depth_hint(gc_ctx *ctx, uint64_t shape) {
    std::lock_guard<std::mutex> lk(ctx->fuse_mu);
    if (!ctx->fuse) return 0;
    auto it = ctx->fuse->depth.find(shape);
    return it == ctx->fuse->depth.end() ? 0 : it->second;
}

uint32_t wg_new(Slot &g, JobRec *j, const CircEntry *ent, bool may_fuse) {
    WgRec w;
    w.head = w.tail = (uint32_t)g.jobs.size();
    w.n = 1;
    w.gates = (uint32_t)ent->gates.size();
    w.slots = ent->job.zslot + 1;
    w.inputs = ent->nin;
    w.depth_sum = ent->circ->plan.p.n_hash_phases;
    w.shape = fuse_shape(0, ent, nullptr);
    w.open = may_fuse;
    j->wg = (uint32_t)g.wgs.size();
    j->member = 0;
    j->next = -1;
    g.wgs.push_back(w);
    return j->wg;
}

uint32_t wg_covering(const Slot &g, const uint32_t *units, uint32_t n) {
    if (n == 0 || n == kDepsAllEarlier) return kFuseMulti;
    uint32_t b = units[0];
    for (uint32_t i = 1; i < n; i++) b = std::max(b, units[i]);
    if (b >= g.wgs.size()) return kFuseMulti;
    const WgRec &w = g.wgs[b];
    for (uint32_t i = 0; i < n; i++)
        if (units[i] != b && !w.behind(units[i])) return kFuseMulti;
    return b;
}

void wg_wait(Slot &g, uint32_t unit, const uint32_t *units, uint32_t n) {
    WgRec &w = g.wgs[unit];
    w.dep_off = (uint32_t)g.unit_deps.size();
    w.dep_n = 0;
    if (n == kDepsAllEarlier) {
        w.dep_n = n;
        for (uint32_t u = 0; u < unit; u++) w.anc[u >> 6] |= 1ull << (u & 63u);
    } else {
        for (uint32_t i = 0; i < n; i++) {
            const WgRec &d = g.wgs[units[i]];
            for (int k = 0; k < 4; k++) w.anc[k] |= d.anc[k];
        }
        for (uint32_t i = 0; i < n; i++) {  // (a unit that another one of the list waits for need not be named)
            bool implied = false;
            for (uint32_t k = 0; k < n && !implied; k++) implied = k != i && g.wgs[units[k]].behind(units[i]);
            if (!implied) {
                g.unit_deps.push_back(units[i]);
                w.dep_n++;
            }
        }
        for (uint32_t i = 0; i < n; i++) w.anc[units[i] >> 6] |= 1ull << (units[i] & 63u);
    }
    g.has_waits = true;
}

void wg_append(Slot &g, uint32_t unit, JobRec *j, const CircEntry *ent, uint32_t n_ext, uint64_t shape) {
    WgRec &w = g.wgs[unit];
    const uint32_t idx = (uint32_t)g.jobs.size();
    g.jobs[w.tail].next = (int32_t)idx;
    w.tail = idx;
    j->member = w.n++;
    w.inputs += n_ext;
    w.depth_sum += ent->circ->plan.p.n_hash_phases;
    w.shape = shape;
    w.gates += (uint32_t)ent->gates.size();
    w.slots += ent->job.zslot + 1;
    j->wg = unit;
    j->next = -1;
}

}  // namespace gcs

extern "C" gc_plan *gc_plan_create_chain(const gc_gate *const *gates, const uint32_t *ngates, const uint32_t *nwires, const uint32_t *nin,
                                         const uint32_t *nout, const uint32_t *const *wiring, uint32_t nsteps, int *status) try {
    int rc = GC_OK;
    gc_plan *pl = nullptr;
    if (!gates || !ngates || !nwires || !nin || !nout || !wiring || nsteps == 0) rc = GC_E_ARG;
    std::vector<ChainStep> steps;
    for (uint32_t k = 0; rc == GC_OK && k < nsteps; k++) {
        if ((!gates[k] && ngates[k]) || (k && nin[k] && !wiring[k])) rc = GC_E_ARG;
        steps.push_back(ChainStep{gates[k], ngates[k], nwires[k], nin[k], nout[k], wiring[k]});
    }
    std::vector<gc_gate> all;
    std::vector<uint32_t> gate_base;
    uint32_t n_ext = 0, n_tmp = 0, n_out = 0;
    if (rc == GC_OK) rc = merge_chain(steps.data(), nsteps, &all, &gate_base, &n_ext, &n_tmp, &n_out);
    if (rc == GC_OK && !(pl = new (std::nothrow) gc_plan)) rc = GC_E_NOMEM;
    if (rc == GC_OK) rc = gc::build_plan(all.data(), (uint32_t)all.size(), n_ext + n_tmp + n_out, n_ext, n_out, &pl->p, false, gate_base.data(), nsteps);
    if (rc != GC_OK) {
        delete pl;
        pl = nullptr;
    }
    if (status) *status = rc;
    return pl;
} catch (...) {
    const int rc__ = gc::on_exception();
    if (status) *status = rc__;
    return nullptr;
}
