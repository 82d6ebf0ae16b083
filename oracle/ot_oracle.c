/*
 * ot_oracle.c — CPU restatement of the IKNP OT-extension expansion, the AES-CTR column
 * PRG, MITCCRH and the COT pad processing (oracle, test infrastructure).
 *
 *   ot/iknp.go:622-645   newPrg / prg / prgLabels (AES-128-CTR, zero IV, stateful stream)
 *   ot/iknp.go:468-511   IKNPReceiver.receive
 *   ot/iknp.go:197-226   IKNPSender.send
 *   ot/iknp.go:647-683   createLabels (bit-matrix transpose)
 *   ot/mitccrh.go:61-128 MITCCRH
 *   ot/cot.go:136-235    COT.Send / COT.Receive pad handling
 *   ot/mul128_generic.go, ot/mul128_ref.go, ot/gf128.go
 *
 * CTR mode restates Go's crypto/cipher.NewCTR (SP 800-38A §6.5 with the whole 16-byte
 * block as a big-endian counter, starting from the IV = 0).
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- PRG: iknp.go:622-645 ---------------------------------------------- */

void orc_prg_init(orc_prg *p, orc_label key) { /* newPrg :622-630 */
    uint8_t kb[16];
    orc_label_get_data(&key, kb);
    orc_aes_init(&p->aes, kb, 16);
    memset(p->ctr, 0, 16);
    p->used = 16;
}

static void prg_refill(orc_prg *p) {
    orc_aes_encrypt(&p->aes, p->ctr, p->ks);
    for (int i = 15; i >= 0; i--) /* 128-bit big-endian increment */
        if (++p->ctr[i]) break;
    p->used = 0;
}

void orc_prg_bytes(orc_prg *p, uint8_t *buf, size_t n) { /* prg :632-637 */
    for (size_t i = 0; i < n; i++) {
        if (p->used == 16) prg_refill(p);
        buf[i] = p->ks[p->used++];
    }
}

void orc_prg_labels(orc_prg *p, orc_label *labels, size_t n) { /* prgLabels :639-645 */
    uint8_t buf[16];
    for (size_t i = 0; i < n; i++) {
        orc_prg_bytes(p, buf, 16);
        orc_label_set_data(&labels[i], buf);
    }
}

/* ---- createLabels: iknp.go:647-683 -------------------------------------- */

void orc_create_labels(orc_label *l, size_t nl, const uint8_t *buf, int w) {
    size_t end = (size_t)w * 8;
    if (end > nl) end = nl;
    for (int row = 0; row < w; row++) {
        orc_label out[8];
        memset(out, 0, sizeof out);
        for (int j = 0; j < 128; j++) {
            uint8_t b = buf[j * w + row];
            uint64_t mask = (uint64_t)1 << (j & 63);
            for (int bit = 0; bit < 8; bit++) {
                if ((b >> bit) & 1) {
                    if (j < 64) out[bit].d0 |= mask;
                    else out[bit].d1 |= mask;
                }
            }
        }
        size_t base = (size_t)row * 8;
        for (int bit = 0; bit < 8; bit++) {
            size_t i = base + (size_t)bit;
            if (i >= end) return;
            l[i] = out[bit];
        }
    }
}

/* ---- IKNP receiver / sender --------------------------------------------- */

void orc_iknp_receiver_init(orc_iknp_receiver *r, const orc_wire base[ORC_IKNP_K]) { /* :347-356 */
    for (int i = 0; i < ORC_IKNP_K; i++) {
        orc_prg_init(&r->g0[i], base[i].l0);
        orc_prg_init(&r->g1[i], base[i].l1);
    }
}

void orc_iknp_sender_init(orc_iknp_sender *s, orc_label delta, const orc_label k0[ORC_IKNP_K]) { /* :117-122 */
    s->delta = delta;
    for (int i = 0; i < ORC_IKNP_K; i++) orc_prg_init(&s->g0[i], k0[i]);
}

size_t orc_iknp_receive(orc_iknp_receiver *r, const uint8_t *b, size_t n, uint8_t *u_out, orc_label *result) {
    /* :472-477 pack the choice bits LSB first */
    size_t blen = (n + 7) / 8;
    uint8_t *bbuf = calloc(blen ? blen : 1, 1);
    for (size_t i = 0; i < n; i++)
        if (b[i]) bbuf[i / 8] |= (uint8_t)(1u << (i % 8));

    uint8_t *chunk = malloc(ORC_IKNP_CHUNK), *out = malloc(ORC_IKNP_CHUNK);
    uint8_t tmp[ORC_IKNP_CHUNK / ORC_IKNP_K];
    size_t written = 0;
    const size_t chunk_rows = (ORC_IKNP_CHUNK / ORC_IKNP_K) * 8;
    for (size_t ofs = 0; ofs < n;) { /* :482-505 */
        size_t rows = chunk_rows;
        if (rows > n - ofs) rows = n - ofs;
        size_t byte_rows = (rows + 7) / 8;
        for (int i = 0; i < ORC_IKNP_K; i++) {
            orc_prg_bytes(&r->g0[i], chunk + (size_t)i * byte_rows, byte_rows);
            orc_prg_bytes(&r->g1[i], tmp, byte_rows);
            for (size_t k = 0; k < byte_rows; k++) {
                tmp[k] ^= chunk[(size_t)i * byte_rows + k];
                if (ofs / 8 + k < blen) tmp[k] ^= bbuf[ofs / 8 + k]; /* xor() truncates to the shorter slice */
            }
            memcpy(out + (size_t)i * byte_rows, tmp, byte_rows);
        }
        memcpy(u_out + written, out, byte_rows * ORC_IKNP_K); /* SendData(out[:byteRows*128]) :499 */
        written += byte_rows * ORC_IKNP_K;
        orc_create_labels(result + ofs, n - ofs, chunk, (int)byte_rows);
        ofs += rows;
    }
    free(bbuf);
    free(chunk);
    free(out);
    return written;
}

size_t orc_iknp_send(orc_iknp_sender *s, const uint8_t *u_in, size_t n, orc_label *result) {
    uint8_t *t = malloc(ORC_IKNP_CHUNK);
    size_t consumed = 0;
    const size_t chunk_rows = (ORC_IKNP_CHUNK / ORC_IKNP_K) * 8;
    for (size_t ofs = 0; ofs < n;) { /* :202-223; the chunk length is what the receiver sent */
        size_t rows = chunk_rows;
        if (rows > n - ofs) rows = n - ofs;
        size_t byte_rows = (rows + 7) / 8;
        const uint8_t *chunk = u_in + consumed;
        for (int i = 0; i < ORC_IKNP_K; i++) {
            orc_prg_bytes(&s->g0[i], t + (size_t)i * byte_rows, byte_rows);
            if (orc_label_bit(&s->delta, i) == 1)
                for (size_t k = 0; k < byte_rows; k++) t[(size_t)i * byte_rows + k] ^= chunk[(size_t)i * byte_rows + k];
        }
        consumed += byte_rows * ORC_IKNP_K;
        orc_create_labels(result + ofs, n - ofs, t, (int)byte_rows);
        ofs += byte_rows * 8;
    }
    free(t);
    return consumed;
}

/* ---- MITCCRH: mitccrh.go:50-128 ------------------------------------------ */

void orc_mitccrh_init(orc_mitccrh *m, orc_label seed, int batch_size) { /* :61-68 */
    m->batch_size = batch_size;
    m->start = seed;
    m->gid = 0;
    m->key_used = batch_size; /* force renew on first use */
}

static void mitccrh_renew(orc_mitccrh *m) { /* :70-89 */
    for (int i = 0; i < m->batch_size; i++) {
        orc_label key = {m->gid, 0};
        m->gid++;
        key.d0 ^= m->start.d0;
        key.d1 ^= m->start.d1;
        uint8_t kb[16];
        orc_label_get_data(&key, kb);
        orc_aes_init(&m->ciphers[i], kb, 16);
    }
    m->key_used = 0;
}

void orc_mitccrh_hash(orc_mitccrh *m, orc_label *blks, int k, int h) { /* :93-128 */
    if (m->key_used == m->batch_size) mitccrh_renew(m);
    for (int i = 0; i < k; i++) {
        const orc_aes *c = &m->ciphers[m->key_used + i];
        for (int j = 0; j < h; j++) {
            int idx = i * h + j;
            uint8_t tmp[16];
            orc_label t;
            orc_label_get_data(&blks[idx], tmp);
            orc_aes_encrypt(c, tmp, tmp);
            orc_label_set_data(&t, tmp);
            blks[idx].d0 ^= t.d0;
            blks[idx].d1 ^= t.d1;
        }
    }
    m->key_used += k;
}

/* ---- COT pads: cot.go:136-235 -------------------------------------------- */

void orc_cot_send_pads(orc_label seed, orc_label delta, const orc_label *data, const orc_wire *wires, size_t n,
                       orc_label *out) {
    orc_mitccrh m;
    orc_mitccrh_init(&m, seed, 8);
    orc_label pad[16];
    memset(pad, 0, sizeof pad);
    size_t w = 0;
    for (size_t i = 0; i < n; i += 8) { /* cot.go:160-181 */
        size_t end = i + 8;
        if (end > n) end = n;
        for (size_t j = i; j < end; j++) {
            pad[2 * (j - i)] = data[j];
            pad[2 * (j - i) + 1] = data[j];
            pad[2 * (j - i) + 1].d0 ^= delta.d0;
            pad[2 * (j - i) + 1].d1 ^= delta.d1;
        }
        orc_mitccrh_hash(&m, pad, 8, 2);
        for (size_t j = i; j < end; j++) {
            pad[2 * (j - i)].d0 ^= wires[j].l0.d0;
            pad[2 * (j - i)].d1 ^= wires[j].l0.d1;
            pad[2 * (j - i) + 1].d0 ^= wires[j].l1.d0;
            pad[2 * (j - i) + 1].d1 ^= wires[j].l1.d1;
        }
        for (size_t j = 0; j < 2 * (end - i); j++) out[w++] = pad[j];
    }
}

/* ---- ROT pads: rot.go:132-202 --------------------------------------------- */

/* ROT.Send (rot.go:156-172): the sender OVERWRITES wires[j] with the two hashed pads of OT j:
 * pad = [data_j, data_j ^ Delta]; mitccrh.Hash(pad, otBatchSize, 2); wires[j] = {pad[0], pad[1]}.  Nothing but the seed
 * travels (rot.go:145-153). */
void orc_rot_send(orc_label seed, orc_label delta, const orc_label *data, size_t n, orc_wire *wires) {
    orc_mitccrh m;
    orc_mitccrh_init(&m, seed, 8);
    orc_label pad[16];
    memset(pad, 0, sizeof pad);
    for (size_t i = 0; i < n; i += 8) { /* rot.go:157-172 */
        size_t end = i + 8;
        if (end > n) end = n;
        for (size_t j = i; j < end; j++) {
            pad[2 * (j - i)] = data[j];
            pad[2 * (j - i) + 1] = data[j];
            pad[2 * (j - i) + 1].d0 ^= delta.d0;
            pad[2 * (j - i) + 1].d1 ^= delta.d1;
        }
        orc_mitccrh_hash(&m, pad, 8, 2);
        for (size_t j = i; j < end; j++) {
            wires[j].l0 = pad[2 * (j - i)];
            wires[j].l1 = pad[2 * (j - i) + 1];
        }
    }
}

/* ROT.Receive (rot.go:194-199): result[j] = the hashed pad of the label IKNP delivered, in place */
void orc_rot_receive(orc_label seed, orc_label *result, size_t n) {
    orc_mitccrh m;
    orc_mitccrh_init(&m, seed, 8);
    orc_label pad[8];
    memset(pad, 0, sizeof pad);
    for (size_t i = 0; i < n; i += 8) {
        size_t end = 8;
        if (end > n - i) end = n - i;
        for (size_t j = 0; j < end; j++) pad[j] = result[i + j]; /* copy(pad, result[i:]) */
        orc_mitccrh_hash(&m, pad, 8, 1);
        for (size_t j = 0; j < end; j++) result[i + j] = pad[j]; /* copy(result[i:], pad) */
    }
}

void orc_cot_receive_unpad(orc_label seed, const uint8_t *flags, const orc_label *sent, orc_label *result, size_t n) {
    orc_mitccrh m;
    orc_mitccrh_init(&m, seed, 8);
    orc_label pad[8];
    memset(pad, 0, sizeof pad);
    for (size_t i = 0; i < n; i += 8) { /* cot.go:203-232 */
        size_t end = 8;
        if (end > n - i) end = n - i;
        for (size_t j = 0; j < end; j++) pad[j] = result[i + j]; /* copy(pad, result[i:]) */
        orc_mitccrh_hash(&m, pad, 8, 1);
        for (size_t j = 0; j < end; j++) {
            orc_label res0 = sent[2 * (i + j)], res1 = sent[2 * (i + j) + 1];
            result[i + j] = flags[i + j] ? res1 : res0;
            result[i + j].d0 ^= pad[j].d0;
            result[i + j].d1 ^= pad[j].d1;
        }
    }
}

/* ---- GF(2^128): mul128_generic.go:9-46, mul128_ref.go:9-36, gf128.go:14-27 */

static void clmul64(uint64_t a, uint64_t b, uint64_t *lo, uint64_t *hi) {
    uint64_t l = 0, h = 0;
    for (int i = 0; i < 64; i++) {
        if ((b >> i) & 1) {
            if (i == 0) l ^= a;
            else {
                l ^= a << i;
                h ^= a >> (64 - i);
            }
        }
    }
    *lo = l;
    *hi = h;
}

void orc_mul128(orc_label a, orc_label b, orc_label *lo, orc_label *hi) {
    uint64_t p00l, p00h, p01l, p01h, p10l, p10h, p11l, p11h;
    clmul64(a.d0, b.d0, &p00l, &p00h);
    clmul64(a.d0, b.d1, &p01l, &p01h);
    clmul64(a.d1, b.d0, &p10l, &p10h);
    clmul64(a.d1, b.d1, &p11l, &p11h);
    uint64_t midl = p01l ^ p10l, midh = p01h ^ p10h;
    lo->d0 = p00l;
    lo->d1 = p00h ^ midl;
    hi->d0 = midh ^ p11l;
    hi->d1 = p11h;
}

void orc_mul128_ref(orc_label a, orc_label b, orc_label *lo, orc_label *hi) {
    uint8_t r[256];
    memset(r, 0, sizeof r);
    for (int i = 0; i < 128; i++) {
        if (!orc_label_bit(&a, i)) continue;
        for (int j = 0; j < 128; j++)
            if (orc_label_bit(&b, j)) r[i + j] ^= 1;
    }
    lo->d0 = lo->d1 = hi->d0 = hi->d1 = 0;
    for (int i = 0; i < 64; i++) {
        lo->d0 |= (uint64_t)r[i] << i;
        lo->d1 |= (uint64_t)r[64 + i] << i;
        hi->d0 |= (uint64_t)r[128 + i] << i;
        hi->d1 |= (uint64_t)r[192 + i] << i;
    }
}

void orc_inner_product(const orc_label *a, const orc_label *b, size_t n, orc_label *r1, orc_label *r2) {
    r1->d0 = r1->d1 = r2->d0 = r2->d1 = 0;
    for (size_t i = 0; i < n; i++) {
        orc_label lo, hi;
        orc_mul128(a[i], b[i], &lo, &hi);
        r1->d0 ^= lo.d0;
        r1->d1 ^= lo.d1;
        r2->d0 ^= hi.d0;
        r2->d1 ^= hi.d1;
    }
}

/* ---- KOS check: iknp.go:138-194, 373-465 ------------------------------------------------------------ */

static void kos_accumulate(orc_prg *chi_prg, const orc_label *v, const uint8_t *bits, size_t n, orc_label *a0,
                           orc_label *a1, orc_label *x) {
    orc_label chi[1024];
    for (size_t i = 0; i < n; i += 1024) { /* :159-168 / :428-447: blocks of len(chi) = 1024 */
        size_t count = n - i < 1024 ? n - i : 1024;
        orc_prg_labels(chi_prg, chi, count);
        orc_label r0, r1;
        orc_inner_product(chi, v + i, count, &r0, &r1);
        a0->d0 ^= r0.d0;
        a0->d1 ^= r0.d1;
        a1->d0 ^= r1.d0;
        a1->d1 ^= r1.d1;
        if (bits && x)
            for (size_t j = 0; j < count; j++)
                if (bits[i + j]) {
                    x->d0 ^= chi[j].d0;
                    x->d1 ^= chi[j].d1;
                }
    }
}

void orc_kos_receiver_tags(orc_label seed2, const orc_label *result, const uint8_t *b, size_t n,
                           const orc_label *choice_vec, const uint8_t *bcv, orc_label *x, orc_label *t0,
                           orc_label *t1) {
    orc_prg chi_prg;
    orc_prg_init(&chi_prg, seed2);
    x->d0 = x->d1 = t0->d0 = t0->d1 = t1->d0 = t1->d1 = 0;
    kos_accumulate(&chi_prg, result, b, n, t0, t1, x);
    kos_accumulate(&chi_prg, choice_vec, bcv, 256, t0, t1, x); /* :450-462 */
}

int orc_kos_sender_check(orc_label seed2, const orc_label *result, size_t n, const orc_label *choice_vec,
                         orc_label delta, orc_label x, orc_label t0, orc_label t1) {
    orc_prg chi_prg;
    orc_prg_init(&chi_prg, seed2);
    orc_label q0 = {0, 0}, q1 = {0, 0};
    kos_accumulate(&chi_prg, result, NULL, n, &q0, &q1, NULL);
    kos_accumulate(&chi_prg, choice_vec, NULL, 256, &q0, &q1, NULL);
    orc_label r0, r1;
    orc_mul128(x, delta, &r0, &r1); /* :186-188 */
    q0.d0 ^= r0.d0;
    q0.d1 ^= r0.d1;
    q1.d0 ^= r1.d0;
    q1.d1 ^= r1.d1;
    return q0.d0 == t0.d0 && q0.d1 == t0.d1 && q1.d0 == t1.d0 && q1.d1 == t1.d1;
}

/* ---- bit-COT: iknp.go:259-310, 554-620 ---------------------------------------------------------------- */

size_t orc_iknp_receive_bits(orc_iknp_receiver *r, const uint64_t *choices, size_t n, uint8_t *u_out, uint64_t *result) {
    uint8_t *chunk = malloc(ORC_IKNP_CHUNK), *tmp = malloc(ORC_IKNP_CHUNK), *ucol = malloc(ORC_IKNP_CHUNK);
    orc_label *labels = malloc(sizeof(orc_label) * 512);
    size_t written = 0;
    for (size_t i = 0; i < (n + 63) / 64; i++) result[i] = 0;
    for (size_t ofs = 0; ofs < n;) {
        size_t rows = 512;
        if (rows > n - ofs) rows = n - ofs;
        size_t byte_rows = (rows + 7) / 8;
        size_t word_offset = ofs / 64, words = byte_rows / 8; /* :583-584: only whole 64-bit words are folded in */
        for (int i = 0; i < ORC_IKNP_K; i++) {
            orc_prg_bytes(&r->g0[i], chunk + (size_t)i * byte_rows, byte_rows);
            orc_prg_bytes(&r->g1[i], tmp, byte_rows);
            for (size_t k = 0; k < byte_rows; k++) tmp[k] ^= chunk[(size_t)i * byte_rows + k];
            for (size_t w = 0; w < words; w++) {
                uint64_t c = choices[word_offset + w];
                for (int k = 0; k < 8; k++) tmp[w * 8 + k] ^= (uint8_t)(c >> (8 * k)); /* little-endian words :593-597 */
            }
            memcpy(ucol + (size_t)i * byte_rows, tmp, byte_rows);
        }
        memcpy(u_out + written, ucol, byte_rows * ORC_IKNP_K);
        written += byte_rows * ORC_IKNP_K;
        memset(labels, 0, sizeof(orc_label) * 512);
        orc_create_labels(labels, 512, chunk, (int)byte_rows);
        for (size_t row = 0; row < rows; row++)
            if (orc_label_bit(&labels[row], 0)) result[(ofs + row) / 64] |= (uint64_t)1 << ((ofs + row) % 64);
        ofs += rows;
    }
    free(chunk);
    free(tmp);
    free(ucol);
    free(labels);
    return written;
}

size_t orc_iknp_send_bits(orc_iknp_sender *s, const uint8_t *u_in, size_t n, uint64_t *result) {
    uint8_t *t = malloc(ORC_IKNP_CHUNK);
    size_t consumed = 0;
    for (size_t i = 0; i < (n + 63) / 64; i++) result[i] = 0;
    for (size_t ofs = 0; ofs < n;) {
        size_t rows = 512;
        if (rows > n - ofs) rows = n - ofs;
        size_t byte_rows = (rows + 7) / 8;
        const uint8_t *chunk = u_in + consumed;
        for (int i = 0; i < ORC_IKNP_K; i++) {
            orc_prg_bytes(&s->g0[i], t + (size_t)i * byte_rows, byte_rows);
            if (orc_label_bit(&s->delta, i) == 1)
                for (size_t k = 0; k < byte_rows; k++) t[(size_t)i * byte_rows + k] ^= chunk[(size_t)i * byte_rows + k];
        }
        consumed += byte_rows * ORC_IKNP_K;
        size_t max_rows = byte_rows * 8; /* :288-292 */
        if (max_rows > n - ofs) max_rows = n - ofs;
        for (size_t row = 0; row < max_rows; row++) /* column 0, no transpose :294-303 */
            if ((t[row / 8] >> (row % 8)) & 1) result[(ofs + row) / 64] |= (uint64_t)1 << ((ofs + row) % 64);
        ofs += max_rows;
    }
    free(t);
    return consumed;
}
