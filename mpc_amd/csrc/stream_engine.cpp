// stream_engine.cpp — C ABI for the streaming garbler (circuit.Streaming, config 5).
//
// Replaces the bodies of NewStreaming (circuit/stream_garble.go:41-75) and Streaming.Garble (:161-192).
// The persistent wire store (stream.wires) stays on the host — it is touched only at circuit
// boundaries; each Garble call (i) resolves the circuit's input wires through in[] (:131-141),
// (ii) garbles the circuit on the device with the SAME kernels as Circuit.Garble (the tweak restarts
// at 0 per circuit, :174, exactly like a fresh Circuit.Garble) and (iii) serialises the gates into the
// caller's buffer in the reference's wire format (:391-446), byte for byte.  Circuits are cached by
// content, so an SSA instruction that repeats re-uses its levelised plan and device buffers.
#include <algorithm>
#include <cstring>
#include <new>
#include <unordered_map>

#include "engine.h"

using namespace gc;

struct gc_stream {
    gc_ctx *ctx = nullptr;
    std::vector<uint8_t> key;
    gc_label r{};
    std::vector<gc_label> l0;     // global wire -> L0 (L1 = L0 ^ R)
    std::vector<gc_label> tmp_l0; // stream.tmp (only outputs of the current circuit are meaningful)
    std::unordered_map<uint64_t, gc_circ *> cache;
};

namespace {

inline uint64_t be64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}
inline void put_be64(uint8_t *p, uint64_t v) {
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (56 - 8 * i));
}
inline void ensure(gc_stream *s, uint32_t max) {  // ensureWires, 64 Ki-wire pages (:95-100)
    if (max < s->l0.size()) return;
    s->l0.resize(((size_t)max / 0x10000 + 1) * 0x10000, gc_label{0, 0});
}

uint64_t circuit_hash(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t nin, uint32_t nout) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) {
        h ^= v;
        h *= 1099511628211ull;
    };
    mix(ngates);
    mix(nwires);
    mix(nin);
    mix(nout);
    for (uint32_t i = 0; i < ngates; i++) {
        mix(((uint64_t)gates[i].in0 << 32) | gates[i].in1);
        mix(((uint64_t)gates[i].out << 8) | gates[i].op);
    }
    return h;
}

}  // namespace

extern "C" {

gc_stream *gc_stream_create(gc_ctx *ctx, const uint8_t *key, size_t keylen, const uint8_t *rnd, size_t rndlen,
                            const uint32_t *inputs, uint32_t ninputs, int *status) {
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
        s->r = gc_label{be64(rnd) | 0x8000000000000000ull, be64(rnd + 8)};  // R.SetS(true)
        uint32_t mx = 0;
        for (uint32_t i = 0; i < ninputs; i++) mx = std::max(mx, inputs[i]);
        ensure(s, mx);
        for (uint32_t i = 0; i < ninputs; i++)
            s->l0[inputs[i]] = gc_label{be64(rnd + 16 * ((size_t)i + 1)), be64(rnd + 16 * ((size_t)i + 1) + 8)};
    }
    if (status) *status = rc;
    return s;
}

void gc_stream_free(gc_stream *s) {
    if (!s) return;
    for (auto &kv : s->cache) gc_circ_free(kv.second);
    delete s;
}

int gc_stream_get_wire(gc_stream *s, uint32_t w, gc_wire *out) {  // Streaming.GetInput (:117-119)
    if (!s || !out || w >= s->l0.size()) return GC_E_ARG;
    out->l0 = s->l0[w];
    out->l1 = gc_label{s->l0[w].d0 ^ s->r.d0, s->l0[w].d1 ^ s->r.d1};
    return GC_OK;
}

int gc_stream_garble(gc_stream *s, const gc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                     uint32_t nin, const uint32_t *out, uint32_t nout, uint8_t *buf, size_t cap, size_t *written) {
    if (!s || (!gates && ngates) || (nin && !in) || (nout && !out) || !buf || !written) return GC_E_ARG;
    if ((uint64_t)nin + nout > nwires) return GC_E_ARG;
    const uint32_t first_tmp = nin, first_out = nwires - nout;
    // initCircuit (:102-114)
    uint32_t mx = 0;
    for (uint32_t i = 0; i < nin; i++) mx = std::max(mx, in[i]);
    for (uint32_t i = 0; i < nout; i++) mx = std::max(mx, out[i]);
    ensure(s, mx);
    // byte size of the serialised circuit is static: 1 + idx bytes + 16 per row
    size_t need = 0;
    for (uint32_t i = 0; i < ngates; i++) {
        const gc_gate &g = gates[i];
        if (g.op > GC_INV) return GC_E_GATE;
        if (g.out < first_tmp) return GC_E_ARG;  // a gate writing an input-mapped wire: not produced by the compiler
        auto index = [&](uint32_t w) { return w < first_tmp ? in[w] : w >= first_out ? out[w - first_out] : w; };
        const uint32_t ai = index(g.in0), bi = g.op == GC_INV ? 0 : index(g.in1), ci = index(g.out);
        const bool shortf = ai <= 0xffff && bi <= 0xffff && ci <= 0xffff;
        const int wc = g.op == GC_INV ? 2 : 3, rows = g.op == GC_AND ? 2 : g.op == GC_OR ? 3 : g.op == GC_INV ? 1 : 0;
        need += 1 + (size_t)(shortf ? 2 : 4) * wc + 16 * (size_t)rows;
    }
    *written = need;
    if (need > cap) return GC_E_ARG;
    if (ngates == 0) return GC_OK;

    // device circuit (cached by content)
    const uint64_t h = circuit_hash(gates, ngates, nwires, nin, nout);
    gc_circ *circ = nullptr;
    auto it = s->cache.find(h);
    if (it != s->cache.end()) circ = it->second;
    else {
        int st = GC_OK;
        circ = gc_circ_load(s->ctx, gates, ngates, nwires, nin, nout, &st);
        if (!circ) return st;
        s->cache.emplace(h, circ);
    }
    const Plan &p = circ->plan.p;
    // input labels through in[] (Get, :131-141)
    std::vector<gc_label> inl(nin);
    for (uint32_t i = 0; i < nin; i++) inl[i] = s->l0[in[i]];
    std::vector<gc_label> slab(std::max<uint32_t>(p.info.slab_rows, 1)), outl(std::max<uint32_t>(nout, 1));
    int rc = gc_garble_labels(circ, s->key.data(), s->key.size(), &s->r, inl.data(), slab.data(), outl.data());
    if (rc != GC_OK) return rc;
    // Set (:143-157): circuit outputs land in the global store
    for (uint32_t j = 0; j < nout; j++) s->l0[out[j]] = outl[j];

    // wire format (:391-446)
    size_t pos = 0;
    for (uint32_t i = 0; i < ngates; i++) {
        const gc_gate &g = gates[i];
        uint32_t ai, bi = 0, ci;
        bool at = false, bt = false, ct = false;
        auto get = [&](uint32_t w, uint32_t &idx, bool &tmp) {
            if (w < first_tmp) idx = in[w];
            else if (w >= first_out) idx = out[w - first_out];
            else {
                idx = w;
                tmp = true;
            }
        };
        if (g.op != GC_INV) get(g.in1, bi, bt);
        get(g.in0, ai, at);
        get(g.out, ci, ct);
        uint8_t op = g.op;
        if (at) op |= 0x80;
        if (bt) op |= 0x40;
        if (ct) op |= 0x20;
        const int wc = g.op == GC_INV ? 2 : 3;
        if (ai <= 0xffff && bi <= 0xffff && ci <= 0xffff) {
            buf[pos++] = op | 0x10;
            buf[pos++] = (uint8_t)(ai >> 8);
            buf[pos++] = (uint8_t)ai;
            if (wc == 3) {
                buf[pos++] = (uint8_t)(bi >> 8);
                buf[pos++] = (uint8_t)bi;
            }
            buf[pos++] = (uint8_t)(ci >> 8);
            buf[pos++] = (uint8_t)ci;
        } else {
            buf[pos++] = op;
            auto p32 = [&](uint32_t v) {
                buf[pos++] = (uint8_t)(v >> 24);
                buf[pos++] = (uint8_t)(v >> 16);
                buf[pos++] = (uint8_t)(v >> 8);
                buf[pos++] = (uint8_t)v;
            };
            p32(ai);
            if (wc == 3) p32(bi);
            p32(ci);
        }
        const uint32_t r0 = p.row_of_gate[i], r1 = p.row_of_gate[i + 1];
        for (uint32_t r = r0; r < r1; r++) {  // Label.Bytes: BE(D0) || BE(D1)
            put_be64(buf + pos, slab[r].d0);
            put_be64(buf + pos + 8, slab[r].d1);
            pos += 16;
        }
    }
    return pos == need ? GC_OK : GC_E_ARG;
}

// ---- streaming evaluator ----------------------------------------------------------------

}  // extern "C"

struct gc_stream_eval {
    gc_ctx *ctx = nullptr;
    std::vector<uint8_t> key;
    std::vector<gc_label> wires;  // StreamEval.wires (global store)
    std::vector<gc_label> tmp;    // StreamEval.tmp
    std::unordered_map<uint64_t, gc_circ *> cache;
};

extern "C" {

gc_stream_eval *gc_stream_eval_create(gc_ctx *ctx, const uint8_t *key, size_t keylen, int *status) {
    int rc = GC_OK;
    AesKey k;
    gc_stream_eval *e = nullptr;
    if (!ctx) rc = GC_E_ARG;
    else if (!key || !aes_expand_key(key, keylen, &k)) rc = GC_E_KEYSIZE;
    else if (!(e = new (std::nothrow) gc_stream_eval)) rc = GC_E_NOMEM;
    if (e) {
        e->ctx = ctx;
        e->key.assign(key, key + keylen);
    }
    if (status) *status = rc;
    return e;
}

void gc_stream_eval_free(gc_stream_eval *e) {
    if (!e) return;
    for (auto &kv : e->cache) gc_circ_free(kv.second);
    delete e;
}

int gc_stream_eval_set_wire(gc_stream_eval *e, uint32_t w, const gc_label *l) {
    if (!e || !l) return GC_E_ARG;
    if (w >= e->wires.size()) e->wires.resize((size_t)w + 1, gc_label{0, 0});
    e->wires[w] = *l;
    return GC_OK;
}

int gc_stream_eval_get_wire(gc_stream_eval *e, uint32_t w, gc_label *l) {
    if (!e || !l || w >= e->wires.size()) return GC_E_ARG;
    *l = e->wires[w];
    return GC_OK;
}

int gc_stream_eval_circuit(gc_stream_eval *e, uint32_t ngates, uint32_t ntmp, uint32_t nwires, const uint8_t *buf,
                           size_t len, size_t *consumed) {
    if (!e || (!buf && len) || !consumed) return GC_E_ARG;
    if (e->wires.size() < nwires) e->wires.resize(nwires, gc_label{0, 0});  // InitCircuit(numWires, numTmpWires)
    if (e->tmp.size() < ntmp) e->tmp.assign(ntmp, gc_label{0, 0});
    *consumed = 0;
    if (ngates == 0) return GC_OK;
    // parse the gate stream (stream_evaluator.go:272-345) into an SSA gate list:
    //   ids [0, nin) = wires read before this circuit writes them (in order of first use), nin + g = output of gate g
    struct Ref { bool tmp; uint32_t idx; };
    std::vector<gc_gate> gates(ngates);
    std::vector<Ref> dst(ngates), inputs;
    std::vector<gc_label> slab;
    std::unordered_map<uint64_t, uint32_t> cur;  // (tmp, idx) -> current id; inputs get 0x80000000|k until numbered
    auto keyof = [](bool t, uint32_t i) { return ((uint64_t)(t ? 1 : 0) << 32) | i; };
    size_t pos = 0;
    for (uint32_t g = 0; g < ngates; g++) {
        if (pos + 1 > len) return GC_E_ROWS;
        uint8_t gop = buf[pos++];
        const bool at = gop & 0x80, bt = gop & 0x40, ct = gop & 0x20, shortf = gop & 0x10;
        gop &= 0x0f;
        if (gop > GC_INV) return GC_E_GATE;  // "invalid operation"
        const int nw = gop == GC_INV ? 2 : 3;
        uint32_t w[3] = {0, 0, 0};
        for (int i = 0; i < nw; i++) {
            const size_t sz = shortf ? 2 : 4;
            if (pos + sz > len) return GC_E_ROWS;
            for (size_t b = 0; b < sz; b++) w[i] = (w[i] << 8) | buf[pos + b];
            pos += sz;
        }
        const uint32_t rows = gop == GC_AND ? 2 : gop == GC_OR ? 3 : gop == GC_INV ? 1 : 0;
        for (uint32_t r = 0; r < rows; r++) {
            if (pos + 16 > len) return GC_E_ROWS;
            slab.push_back(gc_label{be64(buf + pos), be64(buf + pos + 8)});
            pos += 16;
        }
        auto use = [&](bool t, uint32_t idx) -> uint32_t {
            auto it = cur.find(keyof(t, idx));
            if (it != cur.end()) return it->second;
            const uint32_t id = 0x80000000u | (uint32_t)inputs.size();
            inputs.push_back(Ref{t, idx});
            cur.emplace(keyof(t, idx), id);
            return id;
        };
        gates[g].in0 = use(at, w[0]);
        gates[g].in1 = nw == 3 ? use(bt, w[1]) : gates[g].in0;
        gates[g].op = gop;
        gates[g].level = 0;
        dst[g] = Ref{ct, w[nw - 1]};
        cur[keyof(ct, w[nw - 1])] = g;  // gate index for now; renumbered below
        gates[g].out = g;
    }
    const uint32_t nin = (uint32_t)inputs.size();
    for (uint32_t g = 0; g < ngates; g++) {
        auto fix = [&](uint32_t v) { return (v & 0x80000000u) ? (v & 0x7fffffffu) : nin + v; };
        gates[g].in0 = fix(gates[g].in0);
        gates[g].in1 = gates[g].op == GC_INV ? 0 : fix(gates[g].in1);
        gates[g].out = nin + g;
    }
    const uint32_t cw = nin + ngates;
    // device circuit, cached by content
    const uint64_t h = circuit_hash(gates.data(), ngates, cw, nin, 0);
    gc_circ *circ = nullptr;
    auto it = e->cache.find(h);
    if (it != e->cache.end()) circ = it->second;
    else {
        int st = GC_OK;
        circ = gc_circ_load(e->ctx, gates.data(), ngates, cw, nin, 0, &st);
        if (!circ) return st;
        e->cache.emplace(h, circ);
    }
    std::vector<gc_label> wl(cw, gc_label{0, 0});
    for (uint32_t i = 0; i < nin; i++) {
        const Ref &r = inputs[i];
        if (r.tmp) {
            if (r.idx >= e->tmp.size()) return GC_E_ARG;
            wl[i] = e->tmp[r.idx];
        } else {
            if (r.idx >= e->wires.size()) e->wires.resize((size_t)r.idx + 1, gc_label{0, 0});
            wl[i] = e->wires[r.idx];
        }
    }
    int rc = gc_eval(circ, e->key.data(), e->key.size(), 1, wl.data(), nullptr, slab.data(), slab.size(), nullptr);
    if (rc != GC_OK) return rc;
    for (uint32_t g = 0; g < ngates; g++) {  // streaming.Set(cTmp, cIndex, output), in gate order
        const Ref &d = dst[g];
        if (d.tmp) {
            if (d.idx >= e->tmp.size()) return GC_E_ARG;
            e->tmp[d.idx] = wl[nin + g];
        } else {
            if (d.idx >= e->wires.size()) e->wires.resize((size_t)d.idx + 1, gc_label{0, 0});
            e->wires[d.idx] = wl[nin + g];
        }
    }
    *consumed = pos;
    return GC_OK;
}

}  // extern "C"
