// plan.cpp — build the levelised plan (host only; see plan.h).
#include "plan.h"

#include <algorithm>
#include <numeric>

namespace gc {

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

int build_plan(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs, uint32_t noutputs,
               Plan *out) {
    if ((!gates && ngates) || !out) return GC_E_ARG;
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
    uint32_t id = 0, row = 0, max_level = 0;
    for (uint32_t g = 0; g < ngates; g++) {
        const gc_gate &G = gates[g];
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

    // pass 2: stable sort by (level, class)
    std::vector<uint32_t> order(ngates);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        if (p.level_of_gate[a] != p.level_of_gate[b]) return p.level_of_gate[a] < p.level_of_gate[b];
        return op_class(gates[a].op) < op_class(gates[b].op);
    });
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

    // ---- hash-phase schedule + LDS slot allocation -------------------------------------------------
    // key of a value = (a, x): available after hash phase a and x XOR sub-levels following it.
    //   table-producing gate: runs in hash phase max_a(inputs) + 1       -> key (that, 0)
    //   XOR / XNOR gate:      runs in sub-level x+1 of its inputs' max key -> key (a, x+1)
    // Producer ids: < ninputs = input wire, else ninputs + gate (as in pass 1).
    {
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
        std::iota(ford.begin(), ford.end(), 0u);
        std::stable_sort(ford.begin(), ford.end(), [&](uint32_t a, uint32_t b) { return gkey[a] < gkey[b]; });
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

        // stagger chunks: [<= kSHash hash gates][<= kSXor XOR gates]; only meaningful when slots fit 13 bits
        if (!overflow && high < 8192) {
            SChunk sc{0, 0, 0, 0, 0, 0, {0, 0}};
            bool open_ = false;
            auto flush = [&]() {
                if (open_) p.schunks.push_back(sc);
                sc = SChunk{(uint32_t)p.shdescs.size(), 0, 0, 0, (uint32_t)p.sxdescs.size(), 0, {0, 0}};
                open_ = false;
            };
            flush();
            for (uint32_t si = 0; si < p.fsteps.size(); si++) {
                const Step &st = p.fsteps[si];
                if (st.nonfree) {
                    // a hash phase always opens a new chunk; phases wider than kSHash are cut into pieces
                    // (independent gates; the AND / OR / INV grouping is re-counted per piece)
                    for (uint32_t k = 0; k < st.count;) {
                        flush();
                        const uint32_t n = std::min(kSHash, st.count - k);
                        for (uint32_t e = 0; e < n; e++) {
                            const FDesc &d = p.fdescs[st.first + k + e];
                            const uint32_t op = d.row_op >> kOpShift;
                            if (op == GC_AND) sc.n_and++;
                            else if (op == GC_OR) sc.n_or++;
                            else sc.n_inv++;
                            p.shdescs.push_back(d);
                            p.shgslot.push_back(p.fgslot[st.first + k + e]);
                        }
                        open_ = true;
                        k += n;
                    }
                } else {
                    for (uint32_t k = 0; k < st.count; k++) {
                        if (sc.nx == kSXor) flush();
                        const FDesc &d = p.fdescs[st.first + k];
                        XDesc x;
                        x.lin = d.lin;
                        x.lout = (d.lout & 0x1fffu) | ((d.lout & kFStoreGlobal) ? kXStoreGlobal : 0u) |
                                 (((d.row_op >> kOpShift) == GC_XNOR) ? kXXnor : 0u) | ((si & 0xffffu) << 16);
                        p.sxdescs.push_back(x);
                        p.sxgslot.push_back(p.fgslot[st.first + k]);
                        sc.nx++;
                        open_ = true;
                    }
                }
            }
            if (open_) p.schunks.push_back(sc);
        }
        p.info.n_hash_phases = p.n_hash_phases;
        p.info.n_fused_steps = (uint32_t)p.fsteps.size();
        p.info.n_lds_slots = p.n_lds_slots;
    }
    return GC_OK;
}

}  // namespace gc

extern "C" {

gc_plan *gc_plan_create(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                        uint32_t noutputs, int *status) {
    gc_plan *pl = new (std::nothrow) gc_plan;
    int rc = pl ? gc::build_plan(gates, ngates, nwires, ninputs, noutputs, &pl->p) : GC_E_NOMEM;
    if (rc != GC_OK) {
        delete pl;
        pl = nullptr;
    }
    if (status) *status = rc;
    return pl;
}

void gc_plan_free(gc_plan *pl) { delete pl; }

int gc_plan_get_info(const gc_plan *pl, gc_plan_info *out) {
    if (!pl || !out) return GC_E_ARG;
    *out = pl->p.info;
    return GC_OK;
}

int gc_plan_describe(const gc_plan *pl, uint32_t *level_of_gate, uint32_t *tweak_of_gate, uint32_t *row_of_gate,
                     uint32_t *slot_of_gate) {
    if (!pl) return GC_E_ARG;
    const gc::Plan &p = pl->p;
    if (level_of_gate) std::copy(p.level_of_gate.begin(), p.level_of_gate.end(), level_of_gate);
    if (tweak_of_gate) std::copy(p.tweak_of_gate.begin(), p.tweak_of_gate.end(), tweak_of_gate);
    if (row_of_gate) std::copy(p.row_of_gate.begin(), p.row_of_gate.end(), row_of_gate);
    if (slot_of_gate) std::copy(p.slot_of_gate.begin(), p.slot_of_gate.end(), slot_of_gate);
    return GC_OK;
}

}  // extern "C"
