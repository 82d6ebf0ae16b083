// plan.cpp — build the levelised plan (host only; see plan.h).
#include "plan.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>

namespace gc {

namespace {
// developer aid: GC_TRACE=1 prints the planner's laps to stderr
struct PlanLaps {
    bool on = std::getenv("GC_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gc trace] plan: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
}  // namespace

static inline int op_class(uint8_t op) {
    // execution order inside a level: table-producing gates first so that the blocks that need
    // the AES tables are contiguous; XOR/XNOR (free) last
    switch (op) {
    case GC_AND: return 0;
    case GC_OR: return 1;
    case GC_INV: return 2;
    default: return 3;
    }
}


// ---- flattened schedule (see plan.h) -----------------------------------------------------------------
static void symdiff(const std::vector<uint32_t> &a, const std::vector<uint32_t> &b, std::vector<uint32_t> *o) {
    o->clear();
    size_t i = 0, j = 0;
    while (i < a.size() || j < b.size()) {
        if (j == b.size() || (i < a.size() && a[i] < b[j])) o->push_back(a[i++]);
        else if (i == a.size() || b[j] < a[i]) o->push_back(b[j++]);
        else i++, j++;  // x ^ x = 0
    }
}

namespace {
struct OpView {  // build_flat only reads gates[g].op: it runs from the gate list or from the ops kept for a deferred build
    const uint8_t *ops;
    struct G {
        uint8_t op;
    };
    G operator[](uint32_t g) const { return G{ops[g]}; }
};
}  // namespace

// late == false: every table-producing gate runs in the first hash phase its operands allow (as the serial loop would reach
// it).  late == true: in the LAST phase its consumers allow.  The order of execution is free — tweaks and table rows follow the
// gate list, not the schedule — and what it changes is how long labels live: a compiled n-bit multiplier forms its n^2
// partial products from the inputs alone, so "as early as possible" keeps all of them alive (8 500 labels for 128 bits: no
// LDS plan, 630 levels walked through HBM), "as late as possible" forms each row when the adder below it needs it (a few
// hundred).  finish_flat tries the late form when the early one does not fit a workgroup's LDS.
static void build_flat(const OpView gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs, uint32_t noutputs,
                       const std::vector<uint32_t> &src0, const std::vector<uint32_t> &src1,
                       const std::vector<uint32_t> &cur, Plan *out, bool late = false) {
    Plan &p = *out;
    PlanLaps laps;
    const uint32_t NONE = 0xffffffffu;
    const uint32_t nprod = ninputs + ngates;
    std::vector<uint32_t> phase_of;  // late: the hash phase of every table-producing gate
    if (late) {
        std::vector<uint32_t> early(nprod, 0);
        uint32_t last = 0;
        for (uint32_t g = 0; g < ngates; g++) {
            const uint32_t a = std::max(early[src0[g]], early[src1[g]]);
            early[ninputs + g] = op_class(gates[g].op) != 3 ? a + 1 : a;
            last = std::max(last, early[ninputs + g]);
        }
        // latest chunk in which a value may come into being: consumers come later in the gate list, so one backward pass
        std::vector<uint32_t> need_by(nprod, last);
        phase_of.assign(ngates, 0);
        for (uint32_t g = ngates; g-- > 0;) {
            const uint32_t pid = ninputs + g;
            uint32_t by = need_by[pid];
            if (op_class(gates[g].op) != 3) {
                phase_of[g] = std::max(by, early[pid]);  // (by >= early: every consumer sits at least one phase later)
                by = phase_of[g] - 1;
            }
            need_by[src0[g]] = std::min(need_by[src0[g]], by);
            need_by[src1[g]] = std::min(need_by[src1[g]], by);
        }
    }
    // chunk a (= hash phases before the value exists) and XOR round r (0: input / hashed gate) of every producer
    std::vector<uint32_t> A(nprod, 0), Rr(nprod, 0);
    std::vector<uint8_t> is_free(nprod, 0), rpar(ngates, 0);
    std::vector<std::vector<uint32_t>> ex(ngates);  // sorted term list of every XOR gate (and of the block sums behind them)
    std::map<std::vector<uint32_t>, uint32_t> sums;  // block of terms -> the producer that stands for its sum
    const bool two_level = std::getenv("GC_PLAN_NO_TWO_LEVEL") == nullptr;  // (developer switch: the chain of rounds)
    std::vector<uint32_t> t0, t1, tmp;
    // The FULL expansion of an XOR value of a chunk — down to what exists when the chunk begins (and to the chunk's block sums) —
    // with its parity, kept once it has been asked for: the two-level form below needs it for every long list, and working it
    // out by substituting list into list again for every gate made an XOR-only circuit of 128 levels take ten minutes to plan
    // (the synthetic f = 0 row of the bench line: 300 of the default run's 410 s).  Same sets, same plan.
    std::vector<std::vector<uint32_t>> fx(ngates);
    std::vector<uint8_t> fdone(ngates, 0), fp(ngates, 0);
    std::vector<uint32_t> fstack, facc, fnxt, fsingles;
    auto full_of = [&](uint32_t top) {  // top: the producer id of an XOR gate
        const uint32_t nreal = ninputs + ngates;
        fstack.assign(1, top);
        while (!fstack.empty()) {
            const uint32_t cur = fstack.back(), cg = cur - ninputs, ca = A[cur];
            if (fdone[cg]) {
                fstack.pop_back();
                continue;
            }
            auto own_c = [&](uint32_t t) { return t < nreal && is_free[t] && A[t] == ca; };
            bool ready = true;
            for (uint32_t t : ex[cg])
                if (own_c(t) && !fdone[t - ninputs]) fstack.push_back(t), ready = false;
            if (!ready) continue;
            fsingles.clear();
            for (uint32_t t : ex[cg])
                if (!own_c(t)) fsingles.push_back(t);  // (ex[] is sorted: so are these)
            facc.swap(fsingles);
            uint8_t par = rpar[cg];
            for (uint32_t t : ex[cg])
                if (own_c(t)) {
                    symdiff(facc, fx[t - ninputs], &fnxt);
                    facc.swap(fnxt);
                    par ^= fp[t - ninputs];
                }
            fx[cg] = facc;
            fp[cg] = par;
            fdone[cg] = 1;
            fstack.pop_back();
        }
    };
    for (uint32_t g = 0; g < ngates; g++) {
        const uint32_t pid = ninputs + g, s0 = src0[g], s1 = src1[g];
        const uint32_t a = std::max(A[s0], A[s1]);
        if (op_class(gates[g].op) != 3) {
            A[pid] = late ? phase_of[g] : a + 1;
            continue;
        }
        is_free[pid] = 1;
        A[pid] = a;
        auto rnd = [&](uint32_t s) { return (is_free[s] && A[s] == a) ? Rr[s] : 0u; };
        uint32_t r = std::max(std::max(rnd(s0), rnd(s1)), 1u);
        uint8_t par = gates[g].op == GC_XNOR;
        auto terms_of = [&](uint32_t s, std::vector<uint32_t> *t) {
            if (is_free[s] && A[s] == a && Rr[s] == r) {  // same round: substitute its expansion
                *t = ex[s - ninputs];
                par ^= rpar[s - ninputs];
            } else {
                t->assign(1, s);
            }
        };
        terms_of(s0, &t0);
        terms_of(s1, &t1);
        symdiff(t0, t1, &tmp);
        if (tmp.size() > kFlatMaxTerms && two_level) {
            // Two levels instead of a chain of rounds.  The list is expanded all the way down to what exists when the chunk
            // begins (this chunk's own XOR values are replaced by THEIR lists, whatever their round); those terms are summed
            // in blocks of kFlatMaxTerms by XOuts of ROUND 1 that stand for no wire of the circuit — shared by content: the
            // next value of an accumulator chain has the same leading terms — and the value itself is the XOR of the block
            // sums in ROUND 2.  A chain of n values with 2 n terms (the anti-diagonal of an array multiplier: 64 .. 256 values
            // in ONE chunk) then takes 2 rounds, not n / 16.
            const uint32_t nreal = ninputs + ngates;
            auto own = [&](uint32_t t) { return t < nreal && is_free[t] && A[t] == a; };  // an XOR value of this chunk
            std::vector<uint32_t> full, nxt;
            uint8_t fpar = par;
            for (uint32_t t : tmp)
                if (!own(t)) full.push_back(t);  // (tmp is sorted: so is this)
            for (uint32_t t : tmp)
                if (own(t)) {
                    full_of(t);
                    symdiff(full, fx[t - ninputs], &nxt);
                    full.swap(nxt);
                    fpar ^= fp[t - ninputs];
                }
            std::vector<uint32_t> raw, rest;
            for (uint32_t t : full) ((t >= nreal && A[t] == a) ? rest : raw).push_back(t);  // rest: this chunk's block sums
            const size_t nblocks = (raw.size() + kFlatMaxTerms - 1) / kFlatMaxTerms;
            if (rest.size() + nblocks <= kFlatMaxTerms) {
                for (size_t b0 = 0; b0 < raw.size(); b0 += kFlatMaxTerms) {
                    std::vector<uint32_t> blk(raw.begin() + (long)b0, raw.begin() + (long)std::min(raw.size(), b0 + kFlatMaxTerms));
                    if (blk.size() == 1) {
                        rest.push_back(blk[0]);
                        continue;
                    }
                    auto it = sums.find(blk);
                    if (it == sums.end() || A[it->second] > a) {
                        const uint32_t sp = (uint32_t)A.size();
                        A.push_back(a);  // round 1 of THIS chunk: every term exists by then (an earlier chunk's sum is reused as is)
                        Rr.push_back(1);
                        is_free.push_back(1);
                        ex.push_back(blk);
                        rpar.push_back(0);
                        sums[blk] = sp;
                        it = sums.find(blk);
                    }
                    rest.push_back(it->second);
                }
                std::sort(rest.begin(), rest.end());
                // (a block of raw terms may add up to a block sum the list holds already — shared by content —: x ^ x = 0)
                size_t keep = 0;
                for (size_t i = 0; i < rest.size(); i++) {
                    if (i + 1 < rest.size() && rest[i] == rest[i + 1]) i++;
                    else rest[keep++] = rest[i];
                }
                rest.resize(keep);
                tmp.swap(rest);
                par = fpar;
                r = 2;
            }
        }
        if (tmp.size() > kFlatMaxTerms) {  // materialise the operands, continue in the next round
            r += 1;
            par = gates[g].op == GC_XNOR;
            t0.assign(1, s0);
            t1.assign(1, s1);
            symdiff(t0, t1, &tmp);
        }
        Rr[pid] = r;
        rpar[g] = par;
        ex[g] = tmp;
    }
    laps.lap("flat: term lists");
    // what has to exist as a label: operands of hashed gates, circuit outputs, and the terms of those
    const uint32_t nprod2 = (uint32_t)A.size(), ngates2 = (uint32_t)ex.size();  // with the block sums
    std::vector<uint8_t> need(nprod2, 0), is_output(nprod2, 0);
    std::vector<uint32_t> stack;
    for (uint32_t j = 0; j < noutputs; j++) {
        const uint32_t w = nwires - noutputs + j;
        if (cur[w] != NONE) {
            is_output[cur[w]] = 1;
            stack.push_back(cur[w]);
        }
    }
    for (uint32_t g = 0; g < ngates; g++)
        if (!is_free[ninputs + g]) {
            need[ninputs + g] = 1;  // every table-producing gate runs (its rows are part of the output)
            stack.push_back(src0[g]);
            stack.push_back(src1[g]);
        }
    while (!stack.empty()) {
        const uint32_t s = stack.back();
        stack.pop_back();
        if (need[s] && (s < ninputs || !is_free[s])) continue;
        if (need[s]) continue;
        need[s] = 1;
        if (s >= ninputs && is_free[s])
            for (uint32_t t : ex[s - ninputs]) stack.push_back(t);
    }
    // steps: (a, r) in order; r = 0 is the hash phase of chunk a
    std::vector<uint32_t> gl;  // needed gates
    for (uint32_t g = 0; g < ngates2; g++)
        if (need[ninputs + g]) gl.push_back(g);
    auto key_of = [&](uint32_t g) { return ((uint64_t)A[ninputs + g] << 32) | ((uint64_t)Rr[ninputs + g] << 8); };
    auto sub_of = [&](uint32_t g) -> uint64_t {  // order inside a step
        if (!is_free[ninputs + g]) return (uint64_t)op_class(gates[g].op);
        return 0xffu - std::min<uint64_t>(0xfeu, ex[g].size());  // longest term lists first
    };
    {
        // stable order by (key, order inside the step): gl is ascending, so sorting (key, gate) pairs gives what a stable sort
        // by key would — with the keys computed once (the comparator's indirect look-ups were half of the planner's time on
        // a 256-bit multiplier)
        std::vector<std::pair<uint64_t, uint32_t>> keyed(gl.size());
        for (size_t k = 0; k < gl.size(); k++) keyed[k] = {key_of(gl[k]) | sub_of(gl[k]), gl[k]};
        std::sort(keyed.begin(), keyed.end());
        for (size_t k = 0; k < gl.size(); k++) gl[k] = keyed[k].second;
    }
    std::vector<uint32_t> step_first;  // index into gl
    std::vector<uint32_t> step_of(ngates2, 0);
    for (uint32_t k = 0; k < gl.size(); k++) {
        if (k == 0 || key_of(gl[k]) != key_of(gl[k - 1])) step_first.push_back(k);
        step_of[gl[k]] = (uint32_t)step_first.size() - 1;
    }
    const uint32_t nsteps = (uint32_t)step_first.size();
    step_first.push_back((uint32_t)gl.size());
    p.n_flat_steps = nsteps;
    laps.lap("flat: needed + step order");
    // last reader (step + 1; 0 = never read)
    std::vector<uint32_t> last_use(nprod2, 0);
    for (uint32_t g : gl) {
        const uint32_t st = step_of[g] + 1;
        if (is_free[ninputs + g]) {
            for (uint32_t t : ex[g]) last_use[t] = std::max(last_use[t], st);
        } else {
            last_use[src0[g]] = std::max(last_use[src0[g]], st);
            last_use[src1[g]] = std::max(last_use[src1[g]], st);
        }
    }
    // linear scan over the steps; slots freed after step s are reusable from step s + 1
    std::vector<uint32_t> lds_of(nprod2, 0xffffu), free_list;
    std::vector<std::vector<uint32_t>> expire((size_t)nsteps + 2);
    uint32_t high = 0;
    auto take = [&]() {
        if (!free_list.empty()) {
            const uint32_t s = free_list.back();
            free_list.pop_back();
            return s;
        }
        return high++;
    };
    p.fl_in_lds.assign(ninputs, 0xffff);
    for (uint32_t w = 0; w < ninputs; w++) {
        if (last_use[w] == 0) continue;
        const uint32_t s = take();
        lds_of[w] = s;
        expire[last_use[w]].push_back(s);
    }
    for (uint32_t si = 0; si < nsteps; si++) {
        for (uint32_t s : expire[si]) free_list.push_back(s);  // freed after step si - 1
        for (uint32_t k = step_first[si]; k < step_first[si + 1]; k++) {
            const uint32_t pid = ninputs + gl[k];
            const uint32_t s = take();
            lds_of[pid] = s;
            expire[std::max(last_use[pid], si + 1)].push_back(s);
        }
    }
    if (std::getenv("GC_PLAN_DEBUG")) {  // developer aid: live labels after every step of the flattened schedule
        std::vector<int32_t> delta((size_t)nsteps + 2, 0);
        uint32_t nin_live = 0;
        for (uint32_t w = 0; w < ninputs; w++)
            if (last_use[w]) nin_live++, delta[last_use[w]]--;
        for (uint32_t si = 0; si < nsteps; si++)
            for (uint32_t k = step_first[si]; k < step_first[si + 1]; k++) {
                const uint32_t pid = ninputs + gl[k];
                delta[si]++;
                delta[std::max(last_use[pid], si + 1)]--;
            }
        int32_t live = (int32_t)nin_live;
        std::fprintf(stderr, "[plan] %u steps, %u inputs live at start, high-water %u\n", nsteps, nin_live, high);
        for (uint32_t si = 0; si < nsteps; si++) {
            live += delta[si];
            std::fprintf(stderr, "[plan] step %u %s n=%u live=%d\n", si, is_free[ninputs + gl[step_first[si]]] ? "xor " : "hash",
                         step_first[si + 1] - step_first[si], live);
        }
    }
    const uint32_t zslot = high;  // holds the zero label (padding of term lists)
    if (high + 1 >= 0xffffu) {
        p.fl_in_lds.clear();
        return;  // does not fit 16-bit slot numbers: the flat schedule is not offered for this circuit
    }
    p.n_flat_slots = high + 1;
    for (uint32_t w = 0; w < ninputs; w++)
        if (lds_of[w] != 0xffffu) p.fl_in_lds[w] = (uint16_t)lds_of[w];

    laps.lap("flat: slots");
    // units
    std::vector<FDesc> uh;
    std::vector<XOut> uo;
    std::vector<uint32_t> uhg, uog;
    FUnit u{};
    u.xparts = 1;
    auto emit = [&]() {
        if (uh.empty() && uo.empty()) return;
        u.off16 = (uint32_t)(p.fl_prog.size() / 4);
        u.hfirst = (uint32_t)p.fl_hgslot.size();
        u.ofirst = (uint32_t)p.fl_ogslot.size();
        u.nout = (uint32_t)uo.size();
        for (const FDesc &d : uh) {
            p.fl_prog.push_back(d.lin);
            p.fl_prog.push_back(d.lout);
            p.fl_prog.push_back(d.tweak);
            p.fl_prog.push_back(d.row_op);
        }
        u.outs_off16 = (uint32_t)uh.size();
        // Order the terms inside every XOut (XOR commutes) so that the label reads of one ds_read_b128 lane group
        // fall on different bank quarters.  With 4 instances per tile a label row is 64 bytes = a quarter of the
        // 256-byte bank row, selected by (slot mod 4); a wave reads term k of 16 consecutive XOuts at once and the
        // hardware serves the lanes in four groups of four XOuts: {0,3,5,6} {1,2,4,7} {8,11,13,14} {9,10,12,15}
        // (MI355X_MICROARCH.md, LDS).  Random slots collide ~2.1-fold there; a greedy choice brings that down.
        {
            static const uint8_t grp_of[16] = {0, 1, 1, 0, 1, 0, 0, 1, 2, 3, 3, 2, 3, 2, 2, 3};
            for (size_t blk = 0; blk < uo.size(); blk += 16) {
                uint8_t used[4][8] = {};  // [lane group][term position] -> bit mask of quarters already taken
                for (size_t o = blk; o < std::min(uo.size(), blk + 16); o++) {
                    XOut &x = uo[o];
                    uint8_t(&um)[8] = used[grp_of[o - blk]];
                    for (uint32_t k = 0; k < x.n; k++) {
                        uint32_t best = k;
                        for (uint32_t c = k; c < x.n; c++)
                            if (!(um[k] & (1u << (x.t[c] & 3)))) {
                                best = c;
                                break;
                            }
                        std::swap(x.t[k], x.t[best]);
                        um[k] |= (uint8_t)(1u << (x.t[k] & 3));
                    }
                    for (uint32_t k = x.n; k < (x.n > 4 ? 8u : 4u); k++) um[k] |= (uint8_t)(1u << (zslot & 3));
                }
            }
        }
        for (const XOut &x : uo) {  // 6 dwords each
            for (int i = 0; i < 8; i += 2) p.fl_prog.push_back((uint32_t)x.t[i] | ((uint32_t)x.t[i + 1] << 16));
            p.fl_prog.push_back((uint32_t)x.out | ((uint32_t)x.flags << 16));
            p.fl_prog.push_back((uint32_t)x.n);
        }
        while (p.fl_prog.size() & 3) p.fl_prog.push_back(0);
        u.n16 = (uint32_t)(p.fl_prog.size() / 4) - u.off16;
        p.fl_hgslot.insert(p.fl_hgslot.end(), uhg.begin(), uhg.end());
        p.fl_ogslot.insert(p.fl_ogslot.end(), uog.begin(), uog.end());
        p.fl_units.push_back(u);
        p.fl_unit_stride = std::max(p.fl_unit_stride, (u.n16 + 3u) & ~3u);
        uh.clear(), uo.clear(), uhg.clear(), uog.clear();
        u = FUnit{};
        u.xparts = 1;
    };
    for (uint32_t si = 0; si < nsteps; si++) {
        const uint32_t k0 = step_first[si], k1 = step_first[si + 1];
        if (!is_free[ninputs + gl[k0]]) {
            emit();  // a hash phase never shares a unit with what came before it
            for (uint32_t k = k0; k < k1; k++) {
                if (uh.size() == kUHash) emit();
                const uint32_t g = gl[k], pid = ninputs + g;
                FDesc d;
                d.lin = lds_of[src0[g]] | (lds_of[src1[g]] << 16);
                d.lout = lds_of[pid] | (is_output[pid] ? kFStoreGlobal : 0u);
                d.tweak = p.tweak_of_gate[g];
                d.row_op = p.row_of_gate[g] | ((uint32_t)gates[g].op << kOpShift);
                uh.push_back(d);
                uhg.push_back(p.slot_of_gate[g]);
                if (gates[g].op == GC_AND) u.n_and++;
                else if (gates[g].op == GC_OR) u.n_or++;
                else u.n_inv++;
            }
        } else {
            // an XOR round: needs a barrier after whatever produced its terms -> it may share the unit of the
            // hash phase right before it (hash part | barrier | XOR part) but not a unit that already has XOuts
            if (!uo.empty()) emit();
            for (uint32_t k = k0; k < k1; k++) {
                const uint32_t g = gl[k], pid = ninputs + g;
                const uint32_t nt = (uint32_t)ex[g].size();
                // parts: at most 8 terms per lane (kFlatMaxTerms = 32 -> 1, 2 or 4 parts)
                const uint32_t P = nt <= 8 ? 1 : nt <= 16 ? 2 : 4;
                if (uo.size() + (P + 3) > kUOuts) emit();
                XOut dummy{};
                for (int i = 0; i < 8; i++) dummy.t[i] = (uint16_t)zslot;
                dummy.out = (uint16_t)zslot;
                dummy.flags = kXoPart;
                while (uo.size() % P) {  // the leader's index must be a multiple of P (lists are sorted longest first,
                    uo.push_back(dummy);  // so this only pads after a unit break)
                    uog.push_back(0);
                }
                const uint32_t per = (nt + P - 1) / P;
                for (uint32_t part = 0; part < P; part++) {
                    const uint32_t lo = std::min(nt, per * part), hi = std::min(nt, per * (part + 1));
                    XOut x = dummy;
                    x.n = (uint16_t)(hi - lo);
                    for (uint32_t t = lo; t < hi; t++) x.t[t - lo] = (uint16_t)lds_of[ex[g][t]];
                    if (part == 0) {
                        x.out = (uint16_t)lds_of[pid];
                        x.flags = (uint16_t)((is_output[pid] ? kXoStore : 0) | (rpar[g] ? kXoRpar : 0) |
                                             (P == 2 ? kXoJoin2 : P == 4 ? kXoJoin4 : 0));
                    }
                    uo.push_back(x);
                    uog.push_back(part == 0 && g < ngates ? p.slot_of_gate[g] : 0);  // (a block sum is no wire of the circuit)
                }
                u.xparts = std::max(u.xparts, P);
                p.fl_max_parts = std::max(p.fl_max_parts, P);
                p.n_flat_outs++;
                p.n_flat_terms += (uint32_t)ex[g].size();
            }
        }
    }
    emit();
    laps.lap("flat: units");
}

// ---- hash-phase schedule + LDS slot allocation of the level-walking fused kernels (fused_lds_kernels.hip) ------------
// key of a value = (a, x): available after hash phase a and x XOR sub-levels following it.
//   table-producing gate: runs in hash phase max_a(inputs) + 1       -> key (that, 0)
//   XOR / XNOR gate:      runs in sub-level x+1 of its inputs' max key -> key (a, x+1)
// Producer ids: < ninputs = input wire, else ninputs + gate (as in pass 1 of build_plan).  Built with the flattened plan,
// on first demand (finish_flat): ONE instance of a wide circuit, which takes the level launches, needs neither.
static void build_fused(const OpView gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs, uint32_t noutputs,
                        const std::vector<uint32_t> &src0, const std::vector<uint32_t> &src1,
                        const std::vector<uint32_t> &cur, Plan *out) {
    Plan &p = *out;
    const uint32_t NONE = 0xffffffffu;
        const uint32_t nprod = ninputs + ngates;
        std::vector<uint32_t> ka(nprod, 0), kx(nprod, 0);
        std::vector<uint64_t> gkey(ngates);
        for (uint32_t g = 0; g < ngates; g++) {
            const uint32_t s0 = src0[g], s1 = src1[g];
            uint32_t a = ka[s0], x = kx[s0];
            if (ka[s1] > a || (ka[s1] == a && kx[s1] > x)) {
                a = ka[s1];
                x = kx[s1];
            }
            if (op_class(gates[g].op) != 3) {
                a += 1;
                x = 0;
            } else {
                x += 1;
            }
            ka[ninputs + g] = a;
            kx[ninputs + g] = x;
            gkey[g] = ((uint64_t)a << 40) | ((uint64_t)x << 8) | (uint64_t)op_class(gates[g].op);
        }
        std::vector<uint32_t> ford(ngates);
        {  // stable by key = sorted (key, gate) pairs
            std::vector<std::pair<uint64_t, uint32_t>> keyed(ngates);
            for (uint32_t g = 0; g < ngates; g++) keyed[g] = {gkey[g], g};
            std::sort(keyed.begin(), keyed.end());
            for (uint32_t g = 0; g < ngates; g++) ford[g] = keyed[g].second;
        }
        std::vector<uint32_t> step_of_gate(ngates);
        for (uint32_t k = 0; k < ngates;) {
            const uint64_t key = gkey[ford[k]] >> 8;
            Step st{k, 0, 0, 0, 0, 0};
            while (k < ngates && (gkey[ford[k]] >> 8) == key) {
                const uint32_t g = ford[k];
                if (op_class(gates[g].op) != 3) st.nonfree++;
                if (gates[g].op == GC_AND) st.n_and++;
                else if (gates[g].op == GC_OR) st.n_or++;
                else if (gates[g].op == GC_INV) st.n_inv++;
                step_of_gate[g] = (uint32_t)p.fsteps.size();
                st.count++;
                k++;
            }
            if (st.nonfree) p.n_hash_phases++;
            p.fsteps.push_back(st);
        }
        // last step that reads each producer; circuit outputs stay live to the end (they are stored to
        // the global array when produced, so they need no LDS slot beyond their last reader either)
        const uint32_t NEVER = 0;
        std::vector<uint32_t> last_use(nprod, NEVER);  // step index + 1
        for (uint32_t g = 0; g < ngates; g++) {
            const uint32_t st = step_of_gate[g] + 1;
            last_use[src0[g]] = std::max(last_use[src0[g]], st);
            last_use[src1[g]] = std::max(last_use[src1[g]], st);
        }
        std::vector<uint8_t> is_output(nprod, 0);
        for (uint32_t j = 0; j < noutputs; j++) {
            const uint32_t w = nwires - noutputs + j;
            if (cur[w] != NONE) is_output[cur[w]] = 1;
        }
        // linear scan: slots freed after step s are reusable from step s+1
        std::vector<uint32_t> lds_of(nprod, 0xffffu);
        std::vector<uint32_t> free_list;
        uint32_t high = 0;
        std::vector<std::vector<uint32_t>> expire(p.fsteps.size() + 2);
        auto take = [&]() {
            if (!free_list.empty()) {
                uint32_t s = free_list.back();
                free_list.pop_back();
                return s;
            }
            return high++;
        };
        p.in_lds.assign(ninputs, 0xffff);
        bool overflow = false;
        for (uint32_t w = 0; w < ninputs; w++) {
            if (last_use[w] == NEVER) continue;
            const uint32_t s = take();
            lds_of[w] = s;
            if (s >= 0xffff) overflow = true;
            else p.in_lds[w] = (uint16_t)s;
            expire[last_use[w]].push_back(s);  // last_use is step+1: free after that step
        }
        p.fdescs.resize(ngates);
        p.fgslot.resize(ngates);
        for (uint32_t si = 0, k = 0; si < p.fsteps.size(); si++) {
            for (uint32_t s : expire[si]) free_list.push_back(s);  // freed after step si-1
            const Step &st = p.fsteps[si];
            for (uint32_t e = 0; e < st.count; e++, k++) {
                const uint32_t g = ford[k];
                const uint32_t pid = ninputs + g;
                const uint32_t s = take();
                if (s >= 0xffff) overflow = true;
                lds_of[pid] = s;
                const uint32_t lu = std::max(last_use[pid], si + 1);  // at least until its own step ends
                expire[lu].push_back(s);
                FDesc &d = p.fdescs[k];
                d.lin = (lds_of[src0[g]] & 0xffffu) | ((lds_of[src1[g]] & 0xffffu) << 16);
                d.lout = (s & 0xffffu) | (is_output[pid] ? kFStoreGlobal : 0u) | (e == 0 ? kFLevelStart : 0u);
                // XOR/XNOR gates have no tweak: the field carries the step index so that the flat XOR stream
                // of the kernels can tell sub-levels apart without scalar bookkeeping
                d.tweak = op_class(gates[g].op) == 3 ? si : p.tweak_of_gate[g];
                d.row_op = p.row_of_gate[g] | ((uint32_t)gates[g].op << kOpShift);
                p.fgslot[k] = p.slot_of_gate[g];
            }
        }
        p.n_lds_slots = overflow ? 0xffffffffu : high;
        // chunks: cut before every hash phase, and whenever the staging capacity would overflow
        Chunk ch{0, 0, 0, 0};
        for (uint32_t si = 0; si < p.fsteps.size(); si++) {
            const Step &st = p.fsteps[si];
            const bool cut = ch.nsteps && (st.nonfree || ch.ndesc + st.count > kChunkDescs || ch.nsteps == kChunkSteps);
            if (cut) {
                p.fchunks.push_back(ch);
                ch = Chunk{si, 0, st.first, 0};
            }
            ch.nsteps++;
            ch.ndesc += st.count;
        }
        if (ch.nsteps) p.fchunks.push_back(ch);

        p.info.n_hash_phases = p.n_hash_phases;
        p.info.n_fused_steps = (uint32_t)p.fsteps.size();
        p.info.n_lds_slots = p.n_lds_slots;
}

// Host-side self-check of the flattened schedule: evaluates the PLAINTEXT function by walking the exact unit
// program, LDS slot assignment, part joins and global stores the kernels use, with the kernels' parallel semantics
// (inside a hash part / an XOR part every read happens before any write).  A slot recycled too early, a missing
// materialisation or a wrong term list shows up as a wrong output bit — without a GPU.
int simulate_flat(const Plan &p, const uint8_t *in_bits, uint8_t *out_bits) {
    if (p.n_flat_slots == 0xffffffffu) return GC_E_ARG;
    const uint32_t nin = p.info.ninputs;
    std::vector<uint8_t> lds(p.n_flat_slots, 0), glob(p.info.nslots, 0);
    std::vector<uint8_t> poison(p.n_flat_slots, 0);  // 1 = holds a value, 0 = never written (reads of those are errors)
    const uint32_t zslot = p.n_flat_slots - 1;
    poison[zslot] = 1;
    for (uint32_t w = 0; w < nin; w++) {
        glob[w] = in_bits[w] & 1;
        if (p.fl_in_lds[w] != 0xffff) lds[p.fl_in_lds[w]] = glob[w], poison[p.fl_in_lds[w]] = 1;
    }
    for (const FUnit &u : p.fl_units) {
        const uint32_t *img = p.fl_prog.data() + (size_t)u.off16 * 4;
        const uint32_t nh = u.n_and + u.n_or + u.n_inv;
        std::vector<std::pair<uint32_t, uint8_t>> writes;
        for (uint32_t g = 0; g < nh; g++) {
            const uint32_t lin = img[4 * g], lout = img[4 * g + 1], op = img[4 * g + 3] >> kOpShift;
            const uint32_t sa = lin & 0xffffu, sb = lin >> 16;
            if (!poison[sa] || (op != GC_INV && !poison[sb])) return GC_E_WIRE;
            if ((g < u.n_and) != (op == GC_AND) || (g >= u.n_and + u.n_or) != (op == GC_INV)) return GC_E_GATE;
            const uint8_t a = lds[sa], b = lds[sb];
            const uint8_t v = op == GC_AND ? (a & b) : op == GC_OR ? (a | b) : (uint8_t)(a ^ 1);
            writes.emplace_back(lout & 0xffffu, v);
            if (lout & kFStoreGlobal) glob[p.fl_hgslot[u.hfirst + g]] = v;
        }
        for (auto &w : writes) lds[w.first] = w.second, poison[w.first] = 1;
        writes.clear();
        const uint32_t *outs = img + (size_t)u.outs_off16 * 4;
        std::vector<uint8_t> partial(u.nout, 0);
        for (uint32_t o = 0; o < u.nout; o++) {
            const uint32_t *x = outs + 6 * o;
            const uint32_t n = x[5] & 0xffffu;
            uint8_t acc = 0;
            for (uint32_t k = 0; k < (n > 4 ? 8u : 4u); k++) {  // the kernel reads 4 or 8 slots, padding = zero slot
                const uint32_t sl = (x[k / 2] >> (16 * (k & 1))) & 0xffffu;
                if (k < n ? !poison[sl] : sl != zslot) return GC_E_WIRE;
                acc ^= lds[sl];
            }
            partial[o] = acc;
        }
        for (uint32_t o = 0; o < u.nout; o++) {
            const uint32_t *x = outs + 6 * o;
            const uint32_t flags = x[4] >> 16, slot = x[4] & 0xffffu;
            if (flags & kXoPart) continue;
            uint8_t acc = partial[o];
            const uint32_t parts = (flags & kXoJoin4) ? 4u : (flags & kXoJoin2) ? 2u : 1u;
            if (parts > u.xparts || o % parts || o + parts > u.nout) return GC_E_ARG;
            for (uint32_t k = 1; k < parts; k++) {
                if (!((outs[6 * (o + k) + 4] >> 16) & kXoPart)) return GC_E_ARG;
                acc ^= partial[o + k];
            }
            if (flags & kXoRpar) acc ^= 1;  // XNOR: the garbler's "XOR R once" is the plaintext complement
            writes.emplace_back(slot, acc);
            if (flags & kXoStore) glob[p.fl_ogslot[u.ofirst + o]] = acc;
        }
        for (auto &w : writes) lds[w.first] = w.second, poison[w.first] = 1;
    }
    for (uint32_t j = 0; j < p.info.noutputs; j++) out_bits[j] = glob[p.out_slots[j]];
    return GC_OK;
}

bool wide_for_one_instance(const Plan &p, bool eval) {
    const uint64_t passes = eval ? p.passes_eval : p.passes_garble;
    return p.levels.size() >= 2 && passes * 2 >= (uint64_t)p.levels.size() * 5;
}

void finish_flat(Plan *pp) {
    Plan &p = *pp;
    if (p.flat_built) return;
    PlanLaps laps;
    build_fused(OpView{p.lazy_ops.data()}, p.info.ngates, p.info.nwires, p.info.ninputs, p.info.noutputs, p.lazy_src0,
                p.lazy_src1, p.lazy_cur, &p);
    laps.lap("level-walking schedule");
    auto flat = [&](bool late) {
        p.fl_prog.clear(), p.fl_units.clear(), p.fl_hgslot.clear(), p.fl_ogslot.clear(), p.fl_in_lds.clear();
        p.n_flat_slots = 0xffffffffu;
        p.n_flat_outs = p.n_flat_terms = p.n_flat_steps = p.fl_unit_stride = 0;
        build_flat(OpView{p.lazy_ops.data()}, p.info.ngates, p.info.nwires, p.info.ninputs, p.info.noutputs, p.lazy_src0,
                   p.lazy_src1, p.lazy_cur, &p, late);
    };
    // does ONE instance fit a workgroup's LDS beside the AES table (fused_flat_bytes, fused_flat_kernels.hip)?
    auto fits = [&]() {
        return p.n_flat_slots != 0xffffffffu &&
               ((size_t)kFlatStageOff16 + 2 * (size_t)p.fl_unit_stride + p.n_flat_slots + 1) * 16 <= kFlatLdsBytes;
    };
    // (the level-walking schedule just built keeps the same labels alive as the early flattened one, give or take the zero
    // slot: when those cannot fit, the early flattened build is not worth its time — 50 ms for a 256-bit multiplier)
    const bool early_cannot_fit = p.n_lds_slots == 0xffffffffu || ((size_t)kFlatStageOff16 + p.n_lds_slots) * 16 > kFlatLdsBytes;
    const bool may_late = !std::getenv("GC_PLAN_NO_LATE");
    if (std::getenv("GC_PLAN_LATE")) {  // developer aid: the late schedule for every circuit
        flat(true);
        p.flat_late = true;
    } else if (early_cannot_fit && may_late) {
        flat(true);
        p.flat_late = fits();
        if (!p.flat_late) flat(false);  // no LDS plan either way: keep the schedule every other circuit has
    } else {
        flat(false);
        if (!fits() && may_late) {
            flat(true);
            p.flat_late = fits();
            if (!p.flat_late) flat(false);
        }
    }
    p.flat_built = true;
    p.info.n_flat_slots = p.n_flat_slots;
    p.info.n_flat_outs = p.n_flat_outs;
    p.info.n_flat_terms = p.n_flat_terms;
    p.info.n_flat_steps = p.n_flat_steps;
    p.info.n_flat_units = (uint32_t)p.fl_units.size();
    std::vector<uint8_t>().swap(p.lazy_ops);
    std::vector<uint32_t>().swap(p.lazy_src0);
    std::vector<uint32_t>().swap(p.lazy_src1);
    std::vector<uint32_t>().swap(p.lazy_cur);
}

int build_plan(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs, uint32_t noutputs,
               Plan *out, bool defer_flat, const uint32_t *seg_first, uint32_t nseg) {
    if ((!gates && ngates) || !out) return GC_E_ARG;
    const bool tr__ = std::getenv("GC_TRACE") != nullptr;
    auto t__ = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!tr__) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gc trace] plan: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t__).count());
        t__ = n;
    };
    if (ninputs > nwires || noutputs > nwires) return GC_E_ARG;
    if ((uint64_t)ninputs + ngates >= 0xffffffffull) return GC_E_ARG;
    Plan &p = *out;
    p = Plan{};
    p.info.ngates = ngates;
    p.info.nwires = nwires;
    p.info.ninputs = ninputs;
    p.info.noutputs = noutputs;
    p.info.nslots = ninputs + ngates;

    const uint32_t NONE = 0xffffffffu;
    p.level_of_gate.resize(ngates);
    p.tweak_of_gate.resize(ngates);
    p.row_of_gate.resize((size_t)ngates + 1);
    p.slot_of_gate.resize(ngates);
    p.slot_of_wire.assign(nwires, NONE);
    for (uint32_t w = 0; w < ninputs; w++) p.slot_of_wire[w] = w;

    // pass 1 (original order): levels, tweaks, rows, resolve reads to the producing gate.
    // A gate's slot is only known after sorting, so reads are first recorded as
    // "input wire w" (< ninputs) or "gate g" (ninputs + g).
    std::vector<uint32_t> wire_level(nwires, 0);
    std::vector<uint32_t> src0(ngates), src1(ngates);
    std::vector<uint32_t> cur(nwires, NONE);  // producer id of the wire's current value
    for (uint32_t w = 0; w < ninputs; w++) cur[w] = w;
    uint32_t id = 0, row = 0, max_level = 0, seg = 0;
    for (uint32_t g = 0; g < ngates; g++) {
        const gc_gate &G = gates[g];
        while (seg < nseg && seg_first[seg] <= g) {  // the next fused circuit begins: its tweaks start over (stream_garble.go:174)
            if (seg_first[seg] == g) id = 0;
            seg++;
        }
        if (G.op > GC_INV) return GC_E_GATE;
        const bool unary = (G.op == GC_INV);
        if (G.in0 >= nwires || G.out >= nwires || (!unary && G.in1 >= nwires)) return GC_E_WIRE;
        if (cur[G.in0] == NONE || (!unary && cur[G.in1] == NONE)) return GC_E_WIRE;
        uint32_t level = wire_level[G.in0];
        if (!unary) level = std::max(level, wire_level[G.in1]);
        p.level_of_gate[g] = level;
        src0[g] = cur[G.in0];
        src1[g] = unary ? cur[G.in0] : cur[G.in1];
        p.tweak_of_gate[g] = id;
        p.row_of_gate[g] = row;
        switch (G.op) {
        case GC_AND: id += 2; row += 2; p.info.n_and++; break;
        case GC_OR: id += 1; row += 3; p.info.n_or++; break;
        case GC_INV: id += 1; row += 1; p.info.n_inv++; break;
        case GC_XOR: p.info.n_xor++; break;
        default: p.info.n_xnor++; break;
        }
        if (row > kRowMask) return GC_E_ARG;
        wire_level[G.out] = level + 1;
        max_level = std::max(max_level, level + 1);
        cur[G.out] = ninputs + g;
    }
    p.row_of_gate[ngates] = row;
    p.info.slab_rows = row;
    p.info.nlevels = max_level;

    lap("pass 1");
    // pass 2: stable sort by (level, class)
    std::vector<uint32_t> order(ngates);
    {  // counting sort (stable): key = level * 4 + class
        std::vector<uint32_t> start(((size_t)max_level + 1) * 4 + 1, 0);
        auto key_of = [&](uint32_t g) { return (size_t)p.level_of_gate[g] * 4 + (size_t)op_class(gates[g].op); };
        for (uint32_t g = 0; g < ngates; g++) start[key_of(g) + 1]++;
        for (size_t i = 1; i < start.size(); i++) start[i] += start[i - 1];
        for (uint32_t g = 0; g < ngates; g++) order[start[key_of(g)]++] = g;
    }
    lap("sort by level");
    for (uint32_t k = 0; k < ngates; k++) p.slot_of_gate[order[k]] = ninputs + k;
    auto slot_of_src = [&](uint32_t s) { return s < ninputs ? s : p.slot_of_gate[s - ninputs]; };

    p.descs.resize(ngates);
    p.gate_of_desc = order;
    uint32_t width = 0;
    for (uint32_t k = 0; k < ngates;) {
        uint32_t lvl = p.level_of_gate[order[k]];
        Step st{k, 0, 0, 0, 0, 0};
        while (k < ngates && p.level_of_gate[order[k]] == lvl) {
            uint32_t g = order[k];
            GateDesc &d = p.descs[k];
            d.in0 = slot_of_src(src0[g]);
            d.in1 = slot_of_src(src1[g]);
            d.tweak = p.tweak_of_gate[g];
            d.row_op = p.row_of_gate[g] | ((uint32_t)gates[g].op << kOpShift);
            if (op_class(gates[g].op) != 3) st.nonfree++;
            if (gates[g].op == GC_AND) st.n_and++;
            else if (gates[g].op == GC_OR) st.n_or++;
            else if (gates[g].op == GC_INV) st.n_inv++;
            st.count++;
            k++;
        }
        width = std::max(width, st.count);
        p.levels.push_back(st);
    }
    p.info.max_width = width;
    p.info.n_steps = (uint32_t)p.levels.size();

    for (uint32_t w = ninputs; w < nwires; w++)
        if (cur[w] != NONE) p.slot_of_wire[w] = slot_of_src(cur[w]);
    p.out_slots.resize(noutputs);
    for (uint32_t j = 0; j < noutputs; j++) {
        uint32_t s = p.slot_of_wire[nwires - noutputs + j];
        if (s == NONE) return GC_E_WIRE;
        p.out_slots[j] = s;
    }

    lap("descs + levels");
    lap("hash-phase schedule");
    for (const Step &st : p.levels) {  // passes of 1024 lanes per level for ONE instance (kernels.h: level1_passes)
        const uint32_t fr = st.count - st.nonfree;
        p.passes_garble += (((st.n_and + st.n_or) << 2) + (st.n_inv << 1) + fr + 1023) / 1024;
        p.passes_eval += ((st.n_and << 1) + st.n_or + st.n_inv + fr + 1023) / 1024;
    }
    p.lazy_ops.resize(ngates);
    for (uint32_t g = 0; g < ngates; g++) p.lazy_ops[g] = gates[g].op;
    p.lazy_src0.swap(src0);
    p.lazy_src1.swap(src1);
    p.lazy_cur.swap(cur);
    p.info.n_flat_slots = 0xffffffffu;
    lap("lazy copies");
    if (!defer_flat) finish_flat(&p);
    lap("flat");
    return GC_OK;
}

}  // namespace gc

extern "C" {

gc_plan *gc_plan_create(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                        uint32_t noutputs, int *status) try {
    gc_plan *pl = new (std::nothrow) gc_plan;
    int rc = pl ? gc::build_plan(gates, ngates, nwires, ninputs, noutputs, &pl->p) : GC_E_NOMEM;
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

void gc_plan_free(gc_plan *pl) { delete pl; }

int gc_plan_get_info(const gc_plan *pl, gc_plan_info *out) {
    if (!pl || !out) return GC_E_ARG;
    *out = pl->p.info;
    return GC_OK;
}

int gc_plan_simulate(const gc_plan *pl, const uint8_t *in_bits, uint8_t *out_bits) try {
    if (!pl || (!in_bits && pl->p.info.ninputs) || (!out_bits && pl->p.info.noutputs)) return GC_E_ARG;
    return gc::simulate_flat(pl->p, in_bits, out_bits);
} catch (...) {
    return gc::on_exception();
}

// A 64-bit fingerprint (FNV-1a) of everything the kernels execute for this circuit: level steps and descriptors, the
// hash-phase schedule, the flattened unit program with its LDS slots.  Two builds of the planner that give the same
// fingerprint launch the same device work for the circuit — what bench.py ties measured counters to (scripts/profile.sh),
// instead of the text of plan.cpp.
int gc_plan_fingerprint(const gc_plan *pl, uint64_t *fp) try {
    if (!pl || !fp) return GC_E_ARG;
    const gc::Plan &p = pl->p;
    uint64_t h = 14695981039346656037ull;
    auto mix = [&h](const void *data, size_t n) {
        const uint8_t *b = (const uint8_t *)data;
        for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
        h = (h ^ (uint64_t)n) * 1099511628211ull;
    };
    auto vec = [&mix](const auto &v) { mix(v.data(), v.size() * sizeof(v[0])); };
    vec(p.descs), vec(p.levels), vec(p.out_slots);
    vec(p.fdescs), vec(p.fgslot), vec(p.fsteps), vec(p.fchunks), vec(p.in_lds);
    vec(p.fl_prog), vec(p.fl_units), vec(p.fl_hgslot), vec(p.fl_ogslot), vec(p.fl_in_lds);
    const uint32_t tail[8] = {p.n_flat_slots, p.n_flat_outs, p.n_flat_terms, p.n_flat_steps, p.flat_late ? 1u : 0u,
                              p.fl_unit_stride, p.fl_max_parts, p.n_lds_slots};
    mix(tail, sizeof tail);
    *fp = h;
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_plan_describe(const gc_plan *pl, uint32_t *level_of_gate, uint32_t *tweak_of_gate, uint32_t *row_of_gate,
                     uint32_t *slot_of_gate) try {
    if (!pl) return GC_E_ARG;
    const gc::Plan &p = pl->p;
    if (level_of_gate) std::copy(p.level_of_gate.begin(), p.level_of_gate.end(), level_of_gate);
    if (tweak_of_gate) std::copy(p.tweak_of_gate.begin(), p.tweak_of_gate.end(), tweak_of_gate);
    if (row_of_gate) std::copy(p.row_of_gate.begin(), p.row_of_gate.end(), row_of_gate);
    if (slot_of_gate) std::copy(p.slot_of_gate.begin(), p.slot_of_gate.end(), slot_of_gate);
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

}  // extern "C"
