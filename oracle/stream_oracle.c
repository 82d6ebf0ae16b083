/*
 * stream_oracle.c — CPU restatement of the streaming garbler / evaluator (oracle, test infrastructure).
 *
 *   circuit/stream_garble.go:41-75    NewStreaming (R, then one L0 per input wire)
 *   circuit/stream_garble.go:95-157   wire store, initCircuit, Get / Set indirection (in / out / tmp)
 *   circuit/stream_garble.go:161-449  Garble / garbleGate incl. the wire format (:391-446)
 *   circuit/stream_evaluator.go:29-96,271-432   StreamEval store and the per-gate evaluation loop
 * Gate arithmetic is the same as circuit/garble.go (orc_encrypt_half / orc_encrypt / orc_decrypt);
 * the tweak `id` restarts at 0 for every circuit (stream_garble.go:174, stream_evaluator.go:270).
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

struct orc_stream {
    orc_aes alg;
    orc_label r;
    orc_wire *wires; /* global wire store (stream.wires) */
    size_t nwires;
    orc_wire *tmp;
    size_t ntmp;
};

static inline orc_label sx(orc_label a, orc_label b) {
    a.d0 ^= b.d0;
    a.d1 ^= b.d1;
    return a;
}

static int ensure(orc_stream *s, size_t max) { /* ensureWires :95-100 (64 Ki pages) */
    if (max < s->nwires) return 0;
    size_t n = (max / 0x10000 + 1) * 0x10000;
    orc_wire *w = realloc(s->wires, n * sizeof(orc_wire));
    if (!w) return ORC_E_ARG;
    memset(w + s->nwires, 0, (n - s->nwires) * sizeof(orc_wire));
    s->wires = w;
    s->nwires = n;
    return 0;
}

orc_stream *orc_stream_new(const uint8_t *key, size_t keylen, const uint8_t *rnd, size_t rndlen,
                           const uint32_t *inputs, uint32_t ninputs, int *status) {
    int rc = ORC_OK;
    orc_stream *s = calloc(1, sizeof *s);
    if (rndlen < 16) rc = ORC_E_RAND; /* R first (:46-50) */
    if (rc == ORC_OK) {
        orc_label_set_data(&s->r, rnd);
        orc_label_set_s(&s->r, 1);
        rc = orc_aes_init(&s->alg, key, keylen); /* :52 */
    }
    if (rc == ORC_OK) {
        uint32_t mx = 0;
        for (uint32_t i = 0; i < ninputs; i++)
            if (inputs[i] > mx) mx = inputs[i];
        ensure(s, mx);
        for (uint32_t i = 0; i < ninputs && rc == ORC_OK; i++) { /* :67-73 */
            if (16 * ((size_t)i + 2) > rndlen) {
                rc = ORC_E_RAND;
                break;
            }
            orc_wire w;
            orc_label_set_data(&w.l0, rnd + 16 * ((size_t)i + 1));
            w.l1 = sx(w.l0, s->r);
            s->wires[inputs[i]] = w;
        }
    }
    if (rc != ORC_OK) {
        orc_stream_free(s);
        s = NULL;
    }
    if (status) *status = rc;
    return s;
}

void orc_stream_free(orc_stream *s) {
    if (!s) return;
    free(s->wires);
    free(s->tmp);
    free(s);
}

int orc_stream_get(orc_stream *s, uint32_t w, orc_wire *out) {
    if (w >= s->nwires) return ORC_E_ARG;
    *out = s->wires[w];
    return ORC_OK;
}

static inline int idx1(orc_label l) { return orc_label_s(&l) ? 1 : 0; }
static inline int idx2(orc_label a, orc_label b) { return (orc_label_s(&a) ? 2 : 0) | (orc_label_s(&b) ? 1 : 0); }

static inline void put16(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 8);
    p[1] = (uint8_t)v;
}
static inline void put32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

long orc_stream_garble(orc_stream *s, const orc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                       uint32_t nin, const uint32_t *out, uint32_t nout, uint8_t *buf, size_t cap) {
    /* initCircuit :102-114 */
    uint32_t mx = 0;
    for (uint32_t i = 0; i < nin; i++)
        if (in[i] > mx) mx = in[i];
    for (uint32_t i = 0; i < nout; i++)
        if (out[i] > mx) mx = out[i];
    if (ensure(s, mx)) return ORC_E_ARG;
    if (s->ntmp < nwires) {
        free(s->tmp);
        s->tmp = calloc(nwires, sizeof(orc_wire));
        s->ntmp = nwires;
    }
    const uint32_t first_tmp = nin, first_out = nwires - nout;
    uint32_t id = 0; /* :174 */
    size_t pos = 0;
    const orc_label zero = {0, 0};
    for (uint32_t gi = 0; gi < ngates; gi++) {
        const orc_gate *g = &gates[gi];
        orc_wire a, b, c;
        uint32_t ai = 0, bi = 0, ci = 0;
        int at = 0, bt = 0, ct = 0;
        memset(&a, 0, sizeof a);
        memset(&b, 0, sizeof b);
        memset(&c, 0, sizeof c);
        if (g->op > ORC_INV) return ORC_E_GATE;
#define GETW(w, dst, idx, tmpf)                                       \
    do {                                                              \
        if ((w) < first_tmp) { idx = in[(w)]; dst = s->wires[idx]; }  \
        else if ((w) >= first_out) { idx = out[(w)-first_out]; dst = s->wires[idx]; } \
        else { idx = (w); tmpf = 1; dst = s->tmp[(w)]; }              \
    } while (0)
        if (g->op != ORC_INV) GETW(g->in1, b, bi, bt); /* :207-217 */
        GETW(g->in0, a, ai, at);
        orc_label table[4];
        int tstart = 0, tcount = 0, wcount = 3;
        switch (g->op) {
        case ORC_XOR:
            c.l0 = sx(a.l0, b.l0);
            c.l1 = sx(c.l0, s->r);
            break;
        case ORC_XNOR:
            c.l1 = sx(a.l0, b.l0);
            c.l0 = sx(c.l1, s->r);
            break;
        case ORC_AND: { /* :245-286 */
            int pa = orc_label_s(&a.l0), pb = orc_label_s(&b.l0);
            uint32_t j0 = id, j1 = id + 1;
            id += 2;
            orc_label tg = sx(orc_encrypt_half(&s->alg, a.l0, j0), orc_encrypt_half(&s->alg, a.l1, j0));
            if (pb) tg = sx(tg, s->r);
            orc_label wg0 = orc_encrypt_half(&s->alg, a.l0, j0);
            if (pa) wg0 = sx(wg0, tg);
            orc_label te = sx(sx(orc_encrypt_half(&s->alg, b.l0, j1), orc_encrypt_half(&s->alg, b.l1, j1)), a.l0);
            orc_label we0 = orc_encrypt_half(&s->alg, b.l0, j1);
            if (pb) we0 = sx(sx(we0, te), a.l0);
            c.l0 = sx(wg0, we0);
            c.l1 = sx(c.l0, s->r);
            table[0] = tg;
            table[1] = te;
            tcount = 2;
            break;
        }
        case ORC_OR: { /* :303-340 */
            uint32_t t = id++;
            table[idx2(a.l0, b.l0)] = orc_encrypt(&s->alg, a.l0, b.l0, zero, t);
            table[idx2(a.l0, b.l1)] = orc_encrypt(&s->alg, a.l0, b.l1, zero, t);
            table[idx2(a.l1, b.l0)] = orc_encrypt(&s->alg, a.l1, b.l0, zero, t);
            table[idx2(a.l1, b.l1)] = orc_encrypt(&s->alg, a.l1, b.l1, zero, t);
            int l0i = idx2(a.l0, b.l0);
            c.l0 = c.l1 = table[0];
            if (l0i == 0) c.l1 = sx(c.l1, s->r);
            else c.l0 = sx(c.l0, s->r);
            for (int i = 0; i < 4; i++) table[i] = sx(table[i], i == l0i ? c.l0 : c.l1);
            tstart = 1;
            tcount = 3;
            break;
        }
        default: { /* INV :342-375 */
            uint32_t t = id++;
            table[idx1(a.l0)] = orc_encrypt(&s->alg, a.l0, zero, zero, t);
            table[idx1(a.l1)] = orc_encrypt(&s->alg, a.l1, zero, zero, t);
            int l0i = idx1(a.l0);
            c.l0 = c.l1 = table[0];
            if (l0i == 0) c.l0 = sx(c.l0, s->r);
            else c.l1 = sx(c.l1, s->r);
            for (int i = 0; i < 2; i++) table[i] = sx(table[i], i == l0i ? c.l1 : c.l0);
            tstart = 1;
            tcount = 1;
            wcount = 2;
            break;
        }
        }
        /* output :377-389 */
        if (g->out < first_tmp) { ci = in[g->out]; s->wires[ci] = c; }
        else if (g->out >= first_out) { ci = out[g->out - first_out]; s->wires[ci] = c; }
        else { ci = g->out; ct = 1; s->tmp[g->out] = c; }
        /* wire format :391-446 */
        uint8_t op = g->op;
        if (at) op |= 0x80;
        if (bt) op |= 0x40;
        if (ct) op |= 0x20;
        const int shortf = ai <= 0xffff && bi <= 0xffff && ci <= 0xffff;
        size_t need = 1 + (size_t)(shortf ? 2 : 4) * (size_t)wcount + 16 * (size_t)tcount;
        if (pos + need > cap) return ORC_E_ARG;
        if (shortf) {
            buf[pos++] = op | 0x10;
            put16(buf + pos, ai);
            pos += 2;
            if (wcount == 3) { put16(buf + pos, bi); pos += 2; }
            put16(buf + pos, ci);
            pos += 2;
        } else {
            buf[pos++] = op;
            put32(buf + pos, ai);
            pos += 4;
            if (wcount == 3) { put32(buf + pos, bi); pos += 4; }
            put32(buf + pos, ci);
            pos += 4;
        }
        for (int i = 0; i < tcount; i++) {
            orc_label_get_data(&table[tstart + i], buf + pos);
            pos += 16;
        }
    }
    return (long)pos;
}

/* ---- evaluator: stream_evaluator.go:29-96 (store), :271-432 (loop) ------------------------------- */

struct orc_stream_eval {
    orc_aes alg;
    orc_label *wires;
    size_t nwires;
    orc_label *tmp;
    size_t ntmp;
};

orc_stream_eval *orc_stream_eval_new(const uint8_t *key, size_t keylen, int *status) {
    orc_stream_eval *e = calloc(1, sizeof *e);
    int rc = orc_aes_init(&e->alg, key, keylen);
    if (rc) {
        free(e);
        e = NULL;
    }
    if (status) *status = rc;
    return e;
}

void orc_stream_eval_free(orc_stream_eval *e) {
    if (!e) return;
    free(e->wires);
    free(e->tmp);
    free(e);
}

static int eensure(orc_stream_eval *e, size_t n) {
    if (n <= e->nwires) return 0;
    orc_label *w = realloc(e->wires, n * sizeof(orc_label));
    if (!w) return ORC_E_ARG;
    memset(w + e->nwires, 0, (n - e->nwires) * sizeof(orc_label));
    e->wires = w;
    e->nwires = n;
    return 0;
}

int orc_stream_eval_set(orc_stream_eval *e, uint32_t w, orc_label l) {
    if (eensure(e, (size_t)w + 1)) return ORC_E_ARG;
    e->wires[w] = l;
    return ORC_OK;
}

int orc_stream_eval_get(orc_stream_eval *e, uint32_t w, orc_label *l) {
    if (w >= e->nwires) return ORC_E_ARG;
    *l = e->wires[w];
    return ORC_OK;
}

long orc_stream_eval_circuit(orc_stream_eval *e, uint32_t ngates, uint32_t ntmp, uint32_t nwires, const uint8_t *buf,
                             size_t len) {
    if (eensure(e, nwires)) return ORC_E_ARG; /* InitCircuit(numWires, numTmpWires) */
    if (e->ntmp < ntmp) {
        free(e->tmp);
        e->tmp = calloc(ntmp ? ntmp : 1, sizeof(orc_label));
        e->ntmp = ntmp;
    }
    size_t pos = 0;
    uint32_t id = 0;
    const orc_label zero = {0, 0};
    for (uint32_t gi = 0; gi < ngates; gi++) {
        if (pos + 1 > len) return ORC_E_ROWS;
        uint8_t gop = buf[pos++];
        const int at = gop & 0x80, bt = gop & 0x40, ct = gop & 0x20, shortf = gop & 0x10;
        gop &= 0x0f;
        if (gop > ORC_INV) return ORC_E_GATE;
        const int nw = gop == ORC_INV ? 2 : 3;
        uint32_t w[3] = {0, 0, 0};
        for (int i = 0; i < nw; i++) {
            if (shortf) {
                if (pos + 2 > len) return ORC_E_ROWS;
                w[i] = ((uint32_t)buf[pos] << 8) | buf[pos + 1];
                pos += 2;
            } else {
                if (pos + 4 > len) return ORC_E_ROWS;
                w[i] = ((uint32_t)buf[pos] << 24) | ((uint32_t)buf[pos + 1] << 16) | ((uint32_t)buf[pos + 2] << 8) | buf[pos + 3];
                pos += 4;
            }
        }
        const uint32_t ai = w[0], bi = nw == 3 ? w[1] : 0, ci = w[nw - 1];
        const int tcount = gop == ORC_AND ? 2 : gop == ORC_OR ? 3 : gop == ORC_INV ? 1 : 0;
        orc_label garbled[3];
        for (int i = 0; i < tcount; i++) {
            if (pos + 16 > len) return ORC_E_ROWS;
            orc_label_set_data(&garbled[i], buf + pos);
            pos += 16;
        }
#define EGET(t, i) ((t) ? (((i) < e->ntmp) ? e->tmp[(i)] : zero) : (((i) < e->nwires) ? e->wires[(i)] : zero))
        orc_label a = EGET(at, ai), b = zero, c = zero, outl;
        if (nw == 3) b = EGET(bt, bi);
        switch (gop) {
        case ORC_XOR: case ORC_XNOR: outl = sx(a, b); break;
        case ORC_AND: {
            orc_label wg = orc_encrypt_half(&e->alg, a, id);
            if (orc_label_s(&a)) wg = sx(wg, garbled[0]);
            orc_label we = orc_encrypt_half(&e->alg, b, id + 1);
            if (orc_label_s(&b)) we = sx(sx(we, garbled[1]), a);
            id += 2;
            outl = sx(wg, we);
            break;
        }
        case ORC_OR: {
            int index = idx2(a, b);
            if (index > 0) c = garbled[index - 1];
            outl = orc_decrypt(&e->alg, a, b, id++, c);
            break;
        }
        default: {
            int index = idx1(a);
            if (index > 0) c = garbled[0];
            outl = orc_decrypt(&e->alg, a, zero, id++, c);
            break;
        }
        }
        if (ct) {
            if (ci >= e->ntmp) return ORC_E_ARG;
            e->tmp[ci] = outl;
        } else {
            if (eensure(e, (size_t)ci + 1)) return ORC_E_ARG;
            e->wires[ci] = outl;
        }
    }
    return (long)pos;
}
