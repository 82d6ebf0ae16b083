/*
 * gc_oracle.c — CPU restatement of circuit.Garble / circuit.Eval (oracle, test infrastructure).
 *
 * Follows, gate by gate and in the reference's serial order:
 *   ot/label.go:58-166            label arithmetic
 *   circuit/garble.go:27-143      idx / encrypt / decrypt / makeK / encryptHalf
 *   circuit/garble.go:248-308     Circuit.Garble (rand order: R, then L0 per input wire)
 *   circuit/garble.go:311-482     Gate.garbleInto (XOR, XNOR, AND half-gates, OR, INV)
 *   circuit/eval.go:17-115        Circuit.Eval
 *   circuit/computer.go:15-91     Circuit.Compute (plaintext truth)
 *   circuit/circuit.go:206-254    AssignLevels(TargetYao)
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- ot/label.go ------------------------------------------------------- */

static inline uint64_t be64(const uint8_t *p) {
    return ((uint64_t)p[0] << 56) | ((uint64_t)p[1] << 48) | ((uint64_t)p[2] << 40) | ((uint64_t)p[3] << 32) |
           ((uint64_t)p[4] << 24) | ((uint64_t)p[5] << 16) | ((uint64_t)p[6] << 8) | (uint64_t)p[7];
}
static inline void put_be64(uint8_t *p, uint64_t v) {
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (56 - 8 * i));
}

/* label.go:79-83 */
void orc_label_mul2(orc_label *l) {
    l->d0 <<= 1;
    l->d0 |= (l->d1 >> 63);
    l->d1 <<= 1;
}
/* label.go:86-90 */
void orc_label_mul4(orc_label *l) {
    l->d0 <<= 2;
    l->d0 |= (l->d1 >> 62);
    l->d1 <<= 2;
}
/* label.go:70-76 */
void orc_label_set_s(orc_label *l, int set) {
    if (set) l->d0 |= 0x8000000000000000ull;
    else l->d0 &= 0x7fffffffffffffffull;
}
/* label.go:65-67 */
int orc_label_s(const orc_label *l) { return (l->d0 & 0x8000000000000000ull) != 0; }
/* label.go:129-141: bit i<64 comes from D0 (D0 is the LOW limb here) */
unsigned orc_label_bit(const orc_label *l, int i) {
    uint64_t d = (i > 63) ? l->d1 : l->d0;
    return (unsigned)((d >> (i & 63)) & 1);
}
/* label.go:105-108 */
void orc_label_get_data(const orc_label *l, uint8_t out[16]) {
    put_be64(out, l->d0);
    put_be64(out + 8, l->d1);
}
/* label.go:111-114 */
void orc_label_set_data(orc_label *l, const uint8_t in[16]) {
    l->d0 = be64(in);
    l->d1 = be64(in + 8);
}

static inline orc_label lxor(orc_label a, orc_label b) {
    a.d0 ^= b.d0;
    a.d1 ^= b.d1;
    return a;
}

/* ---- circuit/garble.go:20-143 ------------------------------------------ */

static inline int idx_unary(orc_label l0) { return orc_label_s(&l0) ? 1 : 0; }      /* :20-25 */
static inline int idx2(orc_label l0, orc_label l1) {                                  /* :27-38 */
    return (orc_label_s(&l0) ? 2 : 0) | (orc_label_s(&l1) ? 1 : 0);
}

static inline orc_label aes_label(const orc_aes *alg, orc_label k) {
    uint8_t data[16];
    orc_label pi;
    orc_label_get_data(&k, data);
    orc_aes_encrypt(alg, data, data);
    orc_label_set_data(&pi, data);
    return pi;
}

/* makeK: garble.go:74-83  K = 2a ^ 4b ^ t (t in the low 32 bits of D1) */
static inline orc_label make_k(orc_label a, orc_label b, uint32_t t) {
    orc_label_mul2(&a);
    orc_label_mul4(&b);
    a = lxor(a, b);
    a.d1 ^= (uint64_t)t;
    return a;
}

/* encrypt: garble.go:40-55 */
orc_label orc_encrypt(const orc_aes *alg, orc_label a, orc_label b, orc_label c, uint32_t t) {
    orc_label k = make_k(a, b, t);
    orc_label pi = aes_label(alg, k);
    pi = lxor(pi, k);
    pi = lxor(pi, c);
    return pi;
}

/* decrypt: garble.go:57-72 */
orc_label orc_decrypt(const orc_aes *alg, orc_label a, orc_label b, uint32_t t, orc_label c) {
    orc_label k = make_k(a, b, t);
    orc_label crypted = aes_label(alg, k);
    c = lxor(c, crypted);
    c = lxor(c, k);
    return c;
}

/* encryptHalf: garble.go:104-136   H(x,i) = AES(K) ^ K, K = 2x ^ i */
orc_label orc_encrypt_half(const orc_aes *alg, orc_label x, uint32_t i) {
    orc_label k = x;
    orc_label_mul2(&k);
    k.d1 ^= (uint64_t)i;
    orc_label pi = aes_label(alg, k);
    return lxor(pi, k);
}

/* ---- Gate.garbleInto: garble.go:311-482 -------------------------------- */

static int garble_gate(const orc_gate *g, orc_wire *wires, const orc_aes *enc, orc_label r, uint32_t *idp,
                       orc_label table[4], int *start, int *count) {
    orc_wire a, b, c;
    memset(&a, 0, sizeof a);
    memset(&b, 0, sizeof b);
    memset(&c, 0, sizeof c);
    *start = 0;
    *count = 0;

    switch (g->op) { /* :317-328 */
    case ORC_XOR: case ORC_XNOR: case ORC_AND: case ORC_OR:
        b = wires[g->in1];
        /* fallthrough */
    case ORC_INV:
        a = wires[g->in0];
        break;
    default:
        return ORC_E_GATE;
    }

    switch (g->op) {
    case ORC_XOR: { /* :331-340 */
        orc_label l0 = lxor(a.l0, b.l0);
        c.l0 = l0;
        c.l1 = lxor(l0, r);
        break;
    }
    case ORC_XNOR: { /* :342-351 */
        orc_label l0 = lxor(a.l0, b.l0);
        c.l0 = lxor(l0, r);
        c.l1 = l0;
        break;
    }
    case ORC_AND: { /* :353-395 */
        int pa = orc_label_s(&a.l0);
        int pb = orc_label_s(&b.l0);
        uint32_t j0 = *idp, j1 = *idp + 1;
        *idp += 2;
        /* first half gate */
        orc_label tg = orc_encrypt_half(enc, a.l0, j0);
        tg = lxor(tg, orc_encrypt_half(enc, a.l1, j0));
        if (pb) tg = lxor(tg, r);
        orc_label wg0 = orc_encrypt_half(enc, a.l0, j0);
        if (pa) wg0 = lxor(wg0, tg);
        /* second half gate */
        orc_label te = orc_encrypt_half(enc, b.l0, j1);
        te = lxor(te, orc_encrypt_half(enc, b.l1, j1));
        te = lxor(te, a.l0);
        orc_label we0 = orc_encrypt_half(enc, b.l0, j1);
        if (pb) {
            we0 = lxor(we0, te);
            we0 = lxor(we0, a.l0);
        }
        c.l0 = lxor(wg0, we0);
        c.l1 = lxor(c.l0, r);
        table[0] = tg;
        table[1] = te;
        *count = 2;
        break;
    }
    case ORC_OR: { /* :412-444; c is still the zero wire when encrypting */
        uint32_t id = *idp;
        *idp += 1;
        table[idx2(a.l0, b.l0)] = orc_encrypt(enc, a.l0, b.l0, c.l0, id);
        table[idx2(a.l0, b.l1)] = orc_encrypt(enc, a.l0, b.l1, c.l1, id);
        table[idx2(a.l1, b.l0)] = orc_encrypt(enc, a.l1, b.l0, c.l1, id);
        table[idx2(a.l1, b.l1)] = orc_encrypt(enc, a.l1, b.l1, c.l1, id);
        int l0i = idx2(a.l0, b.l0);
        c.l0 = table[0];
        c.l1 = table[0];
        if (l0i == 0) c.l1 = lxor(c.l1, r);
        else c.l0 = lxor(c.l0, r);
        for (int i = 0; i < 4; i++) table[i] = lxor(table[i], (i == l0i) ? c.l0 : c.l1);
        *start = 1;
        *count = 3;
        break;
    }
    case ORC_INV: { /* :446-474 */
        uint32_t id = *idp;
        *idp += 1;
        orc_label zero = {0, 0};
        table[idx_unary(a.l0)] = orc_encrypt(enc, a.l0, zero, c.l1, id);
        table[idx_unary(a.l1)] = orc_encrypt(enc, a.l1, zero, c.l0, id);
        int l0i = idx_unary(a.l0);
        c.l0 = table[0];
        c.l1 = table[0];
        if (l0i == 0) c.l0 = lxor(c.l0, r);
        else c.l1 = lxor(c.l1, r);
        for (int i = 0; i < 2; i++) table[i] = lxor(table[i], (i == l0i) ? c.l1 : c.l0);
        *start = 1;
        *count = 1;
        break;
    }
    }
    wires[g->out] = c;
    return ORC_OK;
}

/* ---- Circuit.Garble: garble.go:248-308 --------------------------------- */

long orc_garble(const orc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs, const uint8_t *key,
                size_t keylen, const uint8_t *rnd, size_t rndlen, orc_label *r_out, orc_wire *wires,
                orc_label *slab, size_t slab_cap, uint32_t *gate_off) {
    (void)nwires;
    size_t pos = 0;
    orc_label r;
    /* R ← NewLabel(rand); R.SetS(true)   :253-258 */
    if (rndlen < 16) return ORC_E_RAND;
    orc_label_set_data(&r, rnd);
    pos = 16;
    orc_label_set_s(&r, 1);

    orc_aes alg; /* aes.NewCipher(key) :260 */
    int err = orc_aes_init(&alg, key, keylen);
    if (err) return err;

    for (uint32_t i = 0; i < ninputs; i++) { /* :271-278, makeLabels :145-159 */
        if (pos + 16 > rndlen) return ORC_E_RAND;
        orc_label l0;
        orc_label_set_data(&l0, rnd + pos);
        pos += 16;
        wires[i].l0 = l0;
        wires[i].l1 = lxor(l0, r);
    }

    uint32_t id = 0;
    size_t slab_off = 0;
    orc_label table[4];
    for (uint32_t i = 0; i < ngates; i++) { /* :285-299 */
        int start, count;
        err = garble_gate(&gates[i], wires, &alg, r, &id, table, &start, &count);
        if (err) return err;
        if (gate_off) gate_off[i] = (uint32_t)slab_off;
        if (count == 0) continue;
        if (slab_off + (size_t)count > slab_cap) return ORC_E_ARG;
        memcpy(slab + slab_off, table + start, (size_t)count * sizeof(orc_label));
        slab_off += (size_t)count;
    }
    if (gate_off) gate_off[ngates] = (uint32_t)slab_off;
    if (r_out) *r_out = r;
    return (long)slab_off;
}

/* ---- Circuit.Eval: eval.go:17-115 --------------------------------------- */

int orc_eval(const orc_gate *gates, uint32_t ngates, uint32_t nwires, const uint8_t *key, size_t keylen,
             orc_label *wires, const orc_label *slab, size_t slab_rows) {
    (void)nwires;
    orc_aes alg;
    int err = orc_aes_init(&alg, key, keylen);
    if (err) return err;
    uint32_t id = 0;
    size_t off = 0;
    const orc_label zero = {0, 0};
    for (uint32_t i = 0; i < ngates; i++) {
        const orc_gate *g = &gates[i];
        orc_label a, b = zero, c = zero, output;
        switch (g->op) { /* :33-44 */
        case ORC_XOR: case ORC_XNOR: case ORC_AND: case ORC_OR:
            a = wires[g->in0];
            b = wires[g->in1];
            break;
        case ORC_INV:
            a = wires[g->in0];
            break;
        default:
            return ORC_E_GATE;
        }
        switch (g->op) {
        case ORC_XOR: case ORC_XNOR: /* :49-51 */
            output = lxor(a, b);
            break;
        case ORC_AND: { /* :53-78 */
            if (off + 2 > slab_rows) return ORC_E_ROWS;
            int sa = orc_label_s(&a), sb = orc_label_s(&b);
            uint32_t j0 = id, j1 = id + 1;
            id += 2;
            orc_label tg = slab[off], te = slab[off + 1];
            off += 2;
            orc_label wg = orc_encrypt_half(&alg, a, j0);
            if (sa) wg = lxor(wg, tg);
            orc_label we = orc_encrypt_half(&alg, b, j1);
            if (sb) {
                we = lxor(we, te);
                we = lxor(we, a);
            }
            output = lxor(wg, we);
            break;
        }
        case ORC_OR: { /* :80-94 */
            if (off + 3 > slab_rows) return ORC_E_ROWS;
            int index = idx2(a, b);
            if (index > 0) c = slab[off + index - 1];
            off += 3;
            output = orc_decrypt(&alg, a, b, id, c);
            id++;
            break;
        }
        default: { /* INV :96-109 */
            if (off + 1 > slab_rows) return ORC_E_ROWS;
            int index = idx_unary(a);
            if (index > 0) c = slab[off];
            off += 1;
            output = orc_decrypt(&alg, a, zero, id, c);
            id++;
            break;
        }
        }
        wires[g->out] = output;
    }
    return ORC_OK;
}

/* ---- Circuit.Compute: computer.go:15-91 --------------------------------- */

int orc_compute(const orc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                const uint8_t *in_bits, uint8_t *wire_bits) {
    memset(wire_bits, 0, nwires);
    for (uint32_t i = 0; i < ninputs; i++) wire_bits[i] = in_bits[i] ? 1 : 0;
    for (uint32_t i = 0; i < ngates; i++) {
        const orc_gate *g = &gates[i];
        uint8_t r;
        switch (g->op) {
        case ORC_XOR: r = wire_bits[g->in0] ^ wire_bits[g->in1]; break;
        case ORC_XNOR: r = (wire_bits[g->in0] ^ wire_bits[g->in1]) == 0; break;
        case ORC_AND: r = wire_bits[g->in0] & wire_bits[g->in1]; break;
        case ORC_OR: r = wire_bits[g->in0] | wire_bits[g->in1]; break;
        case ORC_INV: r = wire_bits[g->in0] == 0; break;
        default: return ORC_E_GATE;
        }
        wire_bits[g->out] = r;
    }
    return ORC_OK;
}

/* ---- AssignLevels(TargetYao): circuit.go:206-254 ------------------------ */

uint32_t orc_assign_levels(orc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t *max_width) {
    uint32_t *levels = calloc(nwires ? nwires : 1, sizeof(uint32_t));
    uint32_t *count = calloc(nwires ? nwires : 1, sizeof(uint32_t));
    uint32_t max = 0;
    for (uint32_t i = 0; i < ngates; i++) {
        orc_gate *g = &gates[i];
        uint32_t level = levels[g->in0];
        if (g->op != ORC_INV) {
            uint32_t l1 = levels[g->in1];
            if (l1 > level) level = l1;
        }
        g->level = level;
        count[level]++;
        level++; /* TargetYao */
        levels[g->out] = level;
        if (level > max) max = level;
    }
    uint32_t mw = 0;
    for (uint32_t i = 0; i < nwires; i++)
        if (count[i] > mw) mw = count[i];
    if (max_width) *max_width = mw;
    free(levels);
    free(count);
    return max;
}

/* ---- table wire format: garbler.go:69-82 / evaluator.go:40-66 ------------------------------------ */

static void put_be32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}
static uint32_t get_be32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
static int rows_of(uint8_t op) { return op == ORC_AND ? 2 : op == ORC_OR ? 3 : op == ORC_INV ? 1 : 0; }

size_t orc_tables_serialize(const orc_gate *gates, uint32_t ngates, const orc_label *slab, uint8_t *out) {
    size_t pos = 0, row = 0;
    put_be32(out, ngates); /* conn.SendUint32(len(garbled.Gates)) */
    pos = 4;
    for (uint32_t i = 0; i < ngates; i++) {
        int n = rows_of(gates[i].op);
        put_be32(out + pos, (uint32_t)n); /* SendUint32(len(data)) */
        pos += 4;
        for (int r = 0; r < n; r++) { /* SendLabel */
            orc_label_get_data(&slab[row++], out + pos);
            pos += 16;
        }
    }
    return pos;
}

long orc_tables_parse(const orc_gate *gates, uint32_t ngates, const uint8_t *in, size_t len, orc_label *slab) {
    if (len < 4 || get_be32(in) != ngates) return ORC_E_ARG; /* "wrong number of gates" */
    size_t pos = 4, row = 0;
    for (uint32_t i = 0; i < ngates; i++) {
        if (pos + 4 > len) return ORC_E_ROWS;
        uint32_t n = get_be32(in + pos);
        pos += 4;
        if ((int)n != rows_of(gates[i].op)) return ORC_E_ROWS; /* surfaces in Eval (eval.go:54-56,86-89) */
        for (uint32_t r = 0; r < n; r++) {
            if (pos + 16 > len) return ORC_E_ROWS;
            orc_label_set_data(&slab[row++], in + pos);
            pos += 16;
        }
    }
    return (long)row;
}

/* ---- CPU baseline: serial garble+eval loop, one instance per iteration --- */

#include <pthread.h>

typedef struct {
    const orc_gate *gates;
    uint32_t ngates, nwires, ninputs, noutputs;
    const uint8_t *key;
    size_t keylen;
    uint32_t reps;
    uint32_t seed;
    int bad;
} bench_arg;

static uint32_t xorshift(uint32_t *s) {
    uint32_t x = *s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return *s = x;
}

static void *bench_thread(void *p) {
    bench_arg *a = p;
    size_t rndlen = 16 * ((size_t)a->ninputs + 1);
    uint8_t *rnd = malloc(rndlen);
    orc_wire *wires = malloc(sizeof(orc_wire) * a->nwires);
    orc_label *ev = malloc(sizeof(orc_label) * a->nwires);
    size_t cap = 3 * (size_t)a->ngates;
    orc_label *slab = malloc(sizeof(orc_label) * cap);
    uint32_t s = a->seed | 1;
    for (uint32_t rep = 0; rep < a->reps; rep++) {
        for (size_t i = 0; i < rndlen; i++) rnd[i] = (uint8_t)xorshift(&s);
        orc_label r;
        long rows = orc_garble(a->gates, a->ngates, a->nwires, a->ninputs, a->key, a->keylen, rnd, rndlen, &r, wires,
                               slab, cap, NULL);
        if (rows < 0) { a->bad = 1; break; }
        /* evaluator's input labels: bit pattern from the PRNG */
        for (uint32_t i = 0; i < a->ninputs; i++) ev[i] = (xorshift(&s) & 1) ? wires[i].l1 : wires[i].l0;
        if (orc_eval(a->gates, a->ngates, a->nwires, a->key, a->keylen, ev, slab, (size_t)rows)) { a->bad = 1; break; }
        for (uint32_t i = a->nwires - a->noutputs; i < a->nwires; i++) {
            int m0 = ev[i].d0 == wires[i].l0.d0 && ev[i].d1 == wires[i].l0.d1;
            int m1 = ev[i].d0 == wires[i].l1.d0 && ev[i].d1 == wires[i].l1.d1;
            if (!m0 && !m1) a->bad = 1;
        }
    }
    free(rnd);
    free(wires);
    free(ev);
    free(slab);
    return NULL;
}

double orc_bench_garble_eval(const orc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                             uint32_t noutputs, const uint8_t *key, size_t keylen, uint32_t reps, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    bench_arg args[256];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        bench_arg a = {gates, ngates, nwires, ninputs, noutputs, key, keylen, reps / (uint32_t)threads +
                       ((uint32_t)t < reps % (uint32_t)threads ? 1u : 0u), 0x9e3779b9u * (uint32_t)(t + 1), 0};
        args[t] = a;
        pthread_create(&th[t], NULL, bench_thread, &args[t]);
    }
    int bad = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        bad |= args[t].bad;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    return bad ? -1.0 : dt;
}
