/* stream_driver.c — a streamed program driven through the C ABI from plain C, the way a cgo host would: no interpreter
 * between the calls.  scripts/bench_stream.py writes the program (its circuits, its steps with their wire bindings, the
 * key and the garbler's random stream) to a file, runs this, and reads back one line of timings and the SHA-256 of the
 * garbler's bytes, which it compares with the oracle-made golden like its own runs.
 *
 * file format (little endian): "GCSP" u32 version=1 | u32 keylen, key | u64 rndlen, rnd | u32 nprim, prim ids |
 *   u32 ncirc { u32 ngates u32 nwires u32 nin u32 nout, gates (gc_gate: 20 bytes each) } |
 *   u32 nsteps { u32 circ, in[nin], out[nout] } | u32 window
 * usage: stream_driver program.bin            -> {"garble_s": .., "eval_s": .., "bytes": .., "sha256": "..", ...}
 *
 * Build: gcc -O2 -I include tools/stream_driver.c -L mpc_amd/csrc -lgcengine -Wl,-rpath,... (oracle/Makefile-style recipe
 * in __graft_entry__.build()). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gcengine.h"

/* ---- SHA-256 (FIPS 180-4), for the byte stream ---- */
typedef struct {
    uint32_t h[8];
    uint64_t len;
    uint8_t buf[64];
    size_t fill;
} sha256_t;
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(sha256_t *s, const uint8_t *p) {
    uint32_t w[64], a, b, c, d, e, f, g, h;
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    a = s->h[0], b = s->h[1], c = s->h[2], d = s->h[3], e = s->h[4], f = s->h[5], g = s->h[6], h = s->h[7];
    for (int i = 0; i < 64; i++) {
        const uint32_t t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        const uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    s->h[0] += a, s->h[1] += b, s->h[2] += c, s->h[3] += d, s->h[4] += e, s->h[5] += f, s->h[6] += g, s->h[7] += h;
}
static void sha_init(sha256_t *s) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(s->h, iv, sizeof iv);
    s->len = 0;
    s->fill = 0;
}
static void sha_update(sha256_t *s, const uint8_t *p, size_t n) {
    s->len += n;
    if (s->fill) {
        const size_t take = n < 64 - s->fill ? n : 64 - s->fill;
        memcpy(s->buf + s->fill, p, take);
        s->fill += take, p += take, n -= take;
        if (s->fill == 64) sha_block(s, s->buf), s->fill = 0;
    }
    for (; n >= 64; p += 64, n -= 64) sha_block(s, p);
    if (n) memcpy(s->buf, p, n), s->fill = n;
}
static void sha_final(sha256_t *s, char hex[65]) {
    const uint64_t bits = s->len * 8;
    uint8_t pad[72] = {0x80};
    const size_t padn = (s->fill < 56 ? 56 : 120) - s->fill;
    for (int i = 0; i < 8; i++) pad[padn + i] = (uint8_t)(bits >> (56 - 8 * i));
    sha_update(s, pad, padn + 8);
    for (int i = 0; i < 8; i++) sprintf(hex + 8 * i, "%08x", s->h[i]);
}

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
#define DIE(...) (fprintf(stderr, __VA_ARGS__), fprintf(stderr, " [%s]\n", gc_last_error()), exit(1))
static void *xmalloc(size_t n, const char *what) {
    void *p = malloc(n ? n : 1);
    if (!p) DIE("stream_driver: out of memory (%zu bytes for %s)", n, what);
    return p;
}
static void *xcalloc(size_t n, size_t sz, const char *what) {
    void *p = calloc(n ? n : 1, sz);
    if (!p) DIE("stream_driver: out of memory (%zu x %zu bytes for %s)", n, sz, what);
    return p;
}
/* A running 64-bit checksum of a byte stream that does not depend on how the stream is cut into pieces (8-byte words, a
 * carry buffer across pieces): the passes that hand the same bytes out another way (deferred copies, views) are checked
 * against the copying pass through it, a few GB/s, instead of against a second kept copy of a 2.4 GB stream. */
typedef struct {
    uint64_t h, len;
    uint8_t buf[8];
    size_t fill;
} sum64_t;
static void sum_init(sum64_t *s) { s->h = 0x9e3779b97f4a7c15ull, s->len = 0, s->fill = 0; }
static inline void sum_word(sum64_t *s, uint64_t w) {
    s->h = (s->h ^ w) * 0xff51afd7ed558ccdull;
    s->h ^= s->h >> 29;
}
static void sum_update(sum64_t *s, const uint8_t *p, size_t n) {
    s->len += n;
    if (s->fill) {
        const size_t take = n < 8 - s->fill ? n : 8 - s->fill;
        memcpy(s->buf + s->fill, p, take);
        s->fill += take, p += take, n -= take;
        if (s->fill < 8) return;
        uint64_t w;
        memcpy(&w, s->buf, 8);
        sum_word(s, w), s->fill = 0;
    }
    for (; n >= 8; p += 8, n -= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        sum_word(s, w);
    }
    if (n) memcpy(s->buf, p, n), s->fill = n;
}
static uint64_t sum_final(sum64_t *s) {
    uint64_t w = 0;
    memcpy(&w, s->buf, s->fill);
    sum_word(s, w);
    sum_word(s, s->len);
    return s->h;
}
static uint64_t sum_of(const uint8_t *p, size_t n) {
    sum64_t s;
    sum_init(&s);
    sum_update(&s, p, n);
    return sum_final(&s);
}
static void rd(FILE *f, void *p, size_t n) {
    if (fread(p, 1, n, f) != n) DIE("stream_driver: short program file");
}
static uint32_t rd32(FILE *f) {
    uint32_t v;
    rd(f, &v, 4);
    return v;
}

typedef struct {
    uint32_t ngates, nwires, nin, nout, handle;
    gc_gate *gates;
} circ_t;
typedef struct {
    uint32_t circ;
    uint32_t *in, *out;
} step_t;

int main(int argc, char **argv) {
    if (argc < 2) DIE("usage: stream_driver program.bin");
    FILE *f = fopen(argv[1], "rb");
    if (!f) DIE("stream_driver: cannot open %s", argv[1]);
    char magic[4];
    rd(f, magic, 4);
    if (memcmp(magic, "GCSP", 4) || rd32(f) != 1) DIE("stream_driver: not a program file");
    const uint32_t keylen = rd32(f);
    uint8_t key[32];
    if (keylen > 32) DIE("stream_driver: key length");
    rd(f, key, keylen);
    uint64_t rndlen;
    rd(f, &rndlen, 8);
    uint8_t *rnd = xmalloc(rndlen, "the random stream");
    rd(f, rnd, rndlen);
    const uint32_t nprim = rd32(f);
    uint32_t *prim = xmalloc(4 * (size_t)nprim + 4, "primary inputs");
    rd(f, prim, 4 * (size_t)nprim);
    const uint32_t ncirc = rd32(f);
    circ_t *circ = xcalloc(ncirc, sizeof *circ, "circuits");
    for (uint32_t c = 0; c < ncirc; c++) {
        circ[c].ngates = rd32(f), circ[c].nwires = rd32(f), circ[c].nin = rd32(f), circ[c].nout = rd32(f);
        circ[c].gates = xmalloc(sizeof(gc_gate) * (size_t)circ[c].ngates, "a circuit's gates");
        rd(f, circ[c].gates, sizeof(gc_gate) * (size_t)circ[c].ngates);
    }
    const uint32_t nsteps = rd32(f);
    step_t *step = xcalloc(nsteps, sizeof *step, "steps");
    size_t cap = 64;
    uint32_t max_wire = 0;
    for (uint32_t k = 0; k < nsteps; k++) {
        const uint32_t c = step[k].circ = rd32(f);
        if (c >= ncirc) DIE("stream_driver: bad circuit index");
        step[k].in = xmalloc(4 * (size_t)circ[c].nin + 4, "a step's inputs"), step[k].out = xmalloc(4 * (size_t)circ[c].nout + 4, "a step's outputs");
        rd(f, step[k].in, 4 * (size_t)circ[c].nin);
        rd(f, step[k].out, 4 * (size_t)circ[c].nout);
        cap += (size_t)circ[c].ngates * 61;
        for (uint32_t i = 0; i < circ[c].nin; i++) max_wire = step[k].in[i] > max_wire ? step[k].in[i] : max_wire;
        for (uint32_t i = 0; i < circ[c].nout; i++) max_wire = step[k].out[i] > max_wire ? step[k].out[i] : max_wire;
    }
    const uint32_t window = rd32(f);
    fclose(f);

    /* eight hardware queues for the engine's streams (the runtime's default is 4; read on the first HIP call of the process):
     * the host's decision, as a Go host would make it in its init() — the library does not touch the environment */
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int st = 0;
    gc_ctx *ctx = gc_ctx_create(0, &st);
    if (!ctx) DIE("gc_ctx_create: %d", st);
    /* (cap is the format's upper bound, 61 bytes per gate; only the ~17 bytes per gate that are written get pages) */
    uint8_t *bytes = xmalloc(cap, "the byte stream");
    size_t *sizes = xmalloc(sizeof(size_t) * (size_t)nsteps, "step sizes");
    double garble_s = 0, eval_s = 0, eval_steady_s = 0;
    double g_t0 = 0, g_t1 = 0, e_t0 = 0, e_t1 = 0; /* CLOCK_MONOTONIC: the same clock in every process of the box */
    uint32_t eval_steady_steps = 0;
    char hex[65] = "";
    gc_label *in0 = xmalloc(sizeof(gc_label) * ((size_t)nprim + 1), "input labels");
    size_t total = 0;
    gc_wire last_w = {{0, 0}, {0, 0}}; /* the garbler's two labels of the program's last output: the evaluator must end on one of them */
    for (int pass = 0; pass < 2; pass++) { /* the first pass loads the circuits and sizes the engine's buffers */
        gc_stream *g = gc_stream_create(ctx, key, keylen, rnd, rndlen, prim, nprim, &st);
        if (!g) DIE("gc_stream_create: %d", st);
        for (uint32_t i = 0; i < nprim; i++) {
            gc_wire w;
            if (gc_stream_get_wire(g, prim[i], &w)) DIE("gc_stream_get_wire");
            in0[i] = w.l0;
        }
        for (uint32_t c = 0; c < ncirc; c++)
            if ((st = gc_stream_intern(g, circ[c].gates, circ[c].ngates, circ[c].nwires, circ[c].nin, circ[c].nout, &circ[c].handle)))
                DIE("gc_stream_intern: %d", st);
        size_t off = 0;
        uint32_t issued = 0;
        const double t0 = now_s();
        for (uint32_t k = 0; k < nsteps; k++) {
            const uint32_t lim = k + window < nsteps ? k + window : nsteps;
            for (; issued < lim; issued++)
                if ((st = gc_stream_garble_begin_h(g, circ[step[issued].circ].handle, step[issued].in, step[issued].out)))
                    DIE("gc_stream_garble_begin_h(step %u): %d", issued, st);
            size_t n = 0;
            if ((st = gc_stream_garble_finish(g, bytes + off, cap - off, &n))) DIE("gc_stream_garble_finish(step %u): %d", k, st);
            sizes[k] = n;
            off += n;
        }
        garble_s = now_s() - t0;
        g_t0 = t0, g_t1 = t0 + garble_s;
        total = off;
        if (gc_stream_get_wire(g, step[nsteps - 1].out[0], &last_w)) DIE("gc_stream_get_wire(last output)");
        gc_stream_free(g);
    }
    /* what the copying pass produced, as a checksum and as SHA-256 (the golden the caller compares): the passes below hand the
     * same bytes out another way and are checked against the checksum — no second copy of the stream is kept */
    const uint64_t want_sum = sum_of(bytes, total);
    uint64_t *step_sum = xmalloc(sizeof(uint64_t) * (size_t)nsteps, "step checksums"); /* (to name the step when a pass differs) */
    for (size_t k = 0, o = 0; k < nsteps; o += sizes[k], k++) step_sum[k] = sum_of(bytes + o, sizes[k]);
    sha256_t sh;
    sha_init(&sh);
    sha_update(&sh, bytes, total);
    sha_final(&sh, hex);
    /* the same with the copies DEFERRED to the stream's copier threads (gc_stream_garble_finish_async + one wait at the end):
     * the bytes land in the same buffer (emptied first), this thread only queues and hands out */
    double garble_async_s = 0;
    {
        gc_stream *g = gc_stream_create(ctx, key, keylen, rnd, rndlen, prim, nprim, &st);
        if (!g) DIE("gc_stream_create: %d", st);
        for (uint32_t c = 0; c < ncirc; c++)
            if ((st = gc_stream_intern(g, circ[c].gates, circ[c].ngates, circ[c].nwires, circ[c].nin, circ[c].nout, &circ[c].handle)))
                DIE("gc_stream_intern: %d", st);
        memset(bytes, 0, total); /* (its pages stay: like those of the buffer of the copying pass) */
        size_t off = 0;
        uint32_t issued = 0;
        const double t0 = now_s();
        for (uint32_t k = 0; k < nsteps; k++) {
            const uint32_t lim = k + window < nsteps ? k + window : nsteps;
            for (; issued < lim; issued++)
                if ((st = gc_stream_garble_begin_h(g, circ[step[issued].circ].handle, step[issued].in, step[issued].out)))
                    DIE("gc_stream_garble_begin_h(step %u): %d", issued, st);
            size_t n = 0;
            if ((st = gc_stream_garble_finish_async(g, bytes + off, total - off, &n))) DIE("gc_stream_garble_finish_async(step %u): %d", k, st);
            if (n != sizes[k]) DIE("gc_stream_garble_finish_async(step %u): %zu bytes, %zu when copied", k, n, sizes[k]);
            off += n;
        }
        if ((st = gc_stream_garble_copies_wait(g))) DIE("gc_stream_garble_copies_wait: %d", st);
        garble_async_s = now_s() - t0;
        if (off != total) DIE("deferred copies: %zu bytes of %zu", off, total);
        if (sum_of(bytes, total) != want_sum) {
            size_t o = 0;
            uint32_t k = 0;
            while (k < nsteps && sum_of(bytes + o, sizes[k]) == step_sum[k]) o += sizes[k], k++;
            DIE("deferred copies: other bytes than the copying finish, first in step %u of %u (%zu bytes at offset %zu)", k, nsteps,
                k < nsteps ? sizes[k] : 0, o);
        }
        gc_stream_free(g);
    }
    /* the same program with the bytes consumed IN PLACE (gc_stream_garble_finish_view: a pointer into the engine's pinned staging,
     * valid until the next finish — what go/circuit/stream_hip.go does: the copy into conn.WriteBuf is the transport's).  Twice:
     * timed without touching the bytes (a transport would DMA them), then once more reading every view into the checksum. */
    double garble_view_s = 0;
    for (int pass = 0; pass < 2; pass++) {
        gc_stream *g = gc_stream_create(ctx, key, keylen, rnd, rndlen, prim, nprim, &st);
        if (!g) DIE("gc_stream_create: %d", st);
        for (uint32_t c = 0; c < ncirc; c++)
            if ((st = gc_stream_intern(g, circ[c].gates, circ[c].ngates, circ[c].nwires, circ[c].nin, circ[c].nout, &circ[c].handle)))
                DIE("gc_stream_intern: %d", st);
        uint32_t issued = 0;
        size_t seen = 0;
        sum64_t vs;
        sum_init(&vs);
        const double t0 = now_s();
        for (uint32_t k = 0; k < nsteps; k++) {
            const uint32_t lim = k + window < nsteps ? k + window : nsteps;
            for (; issued < lim; issued++)
                if ((st = gc_stream_garble_begin_h(g, circ[step[issued].circ].handle, step[issued].in, step[issued].out)))
                    DIE("gc_stream_garble_begin_h(step %u): %d", issued, st);
            const uint8_t *view = NULL;
            size_t n = 0;
            if ((st = gc_stream_garble_finish_view(g, &view, &n))) DIE("gc_stream_garble_finish_view(step %u): %d", k, st);
            if (n != sizes[k]) DIE("gc_stream_garble_finish_view(step %u): %zu bytes, %zu when copied", k, n, sizes[k]);
            if (pass == 1) {
                if (sum_of(view, n) != step_sum[k]) DIE("view pass: step %u of %u: other bytes than the copying finish (%zu bytes)", k, nsteps, n);
                sum_update(&vs, view, n);
            }
            seen += n;
        }
        if (pass == 0) garble_view_s = now_s() - t0;
        if (seen != total) DIE("view pass: %zu bytes of %zu", seen, total);
        if (pass == 1 && sum_final(&vs) != want_sum) DIE("view pass: other bytes than the copying finish (checksum of %zu bytes)", total);
        gc_stream_free(g);
    }
    uint64_t parsed = 0, matched = 0;
    gc_label probe = {0, 0};
    for (int pass = 0; pass < 2; pass++) {
        gc_stream_eval *e = gc_stream_eval_create(ctx, key, keylen, &st);
        if (!e) DIE("gc_stream_eval_create: %d", st);
        for (uint32_t i = 0; i < nprim; i++)
            if (gc_stream_eval_set_wire(e, prim[i], &in0[i])) DIE("gc_stream_eval_set_wire");
        size_t off = 0;
        const double t0 = now_s();
        /* the blocks the evaluator sees for the first time are parsed gate by gate and their circuits loaded (the peer's
         * data: nothing can be interned ahead, as the garbler does); "steady" is what follows the last such block */
        double t_known = t0;
        uint64_t parsed_before = 0;
        eval_steady_steps = nsteps;
        for (uint32_t k = 0; k < nsteps; k++) {
            const circ_t *c = &circ[step[k].circ];
            size_t used = 0;
            if ((st = gc_stream_eval_circuit(e, c->ngates, c->nwires, max_wire + 1, bytes + off, sizes[k], &used)) || used != sizes[k])
                DIE("gc_stream_eval_circuit(step %u): %d, used %zu of %zu", k, st, used, sizes[k]);
            off += used;
            gc_stream_eval_stats(e, &parsed, &matched);
            if (parsed != parsed_before) {
                parsed_before = parsed;
                t_known = now_s();
                eval_steady_steps = nsteps - 1 - k;
            }
        }
        if (gc_stream_eval_get_wire(e, step[nsteps - 1].out[0], &probe)) DIE("gc_stream_eval_get_wire"); /* waits for everything */
        eval_s = now_s() - t0;
        if (!((probe.d0 == last_w.l0.d0 && probe.d1 == last_w.l0.d1) || (probe.d0 == last_w.l1.d0 && probe.d1 == last_w.l1.d1)))
            DIE("evaluator: the last output's label %016llx%016llx is neither of the garbler's two", (unsigned long long)probe.d0,
                (unsigned long long)probe.d1);
        e_t0 = t0, e_t1 = t0 + eval_s;
        eval_steady_s = now_s() - t_known;
        gc_stream_eval_stats(e, &parsed, &matched);
        gc_stream_eval_free(e);
    }
    /* The same stream as the peer frames it (compiler/ssa/streamer.go:679-693: OpCircuit, step, numGates, numTmpWires, numWires
     * in front of every block), handed to gc_stream_eval_blocks in pieces of a p2p.Conn read buffer (1 MiB, p2p/protocol.go:25;
     * GC_DRIVER_CHUNK bytes): what a Go host does with conn.ReadBuf[ReadStart:ReadEnd] instead of collecting a block gate by
     * gate.  A piece that ends inside a block is followed by one that starts at that block (a real reader moves the rest to
     * the front of its buffer and reads on). */
    double eval_blocks_s = 0, eval_blocks_pinned_s = 0;
    size_t chunk = (size_t)1 << 20;
    if (getenv("GC_DRIVER_CHUNK")) chunk = (size_t)strtoull(getenv("GC_DRIVER_CHUNK"), NULL, 0);
    if (chunk < 64) chunk = 64;
    {
        const size_t ftotal = total + 20 * (size_t)nsteps;
        uint8_t *framed = xmalloc(ftotal, "the framed stream");
        size_t fo = 0, off = 0;
        for (uint32_t k = 0; k < nsteps; k++) {
            const circ_t *c = &circ[step[k].circ];
            const uint32_t hdr[5] = {1u /* OpCircuit */, k, c->ngates, c->nwires, max_wire + 1};
            for (int i = 0; i < 5; i++) {
                framed[fo++] = (uint8_t)(hdr[i] >> 24), framed[fo++] = (uint8_t)(hdr[i] >> 16);
                framed[fo++] = (uint8_t)(hdr[i] >> 8), framed[fo++] = (uint8_t)hdr[i];
            }
            memcpy(framed + fo, bytes + off, sizes[k]);
            fo += sizes[k], off += sizes[k];
        }
        free(bytes), bytes = NULL; /* (from here on the stream exists once: framed — then once more, pinned, with `framed` given back) */
        for (int pass = 0; pass < 2; pass++) {
            gc_stream_eval *e = gc_stream_eval_create(ctx, key, keylen, &st);
            if (!e) DIE("gc_stream_eval_create: %d", st);
            for (uint32_t i = 0; i < nprim; i++)
                if (gc_stream_eval_set_wire(e, prim[i], &in0[i])) DIE("gc_stream_eval_set_wire");
            const double t0 = now_s();
            size_t pos = 0, win = chunk;
            uint64_t done = 0;
            while (pos < ftotal) {
                const size_t n = ftotal - pos < win ? ftotal - pos : win;
                size_t used = 0;
                uint32_t nb = 0;
                int more = 0;
                if ((st = gc_stream_eval_blocks(e, framed + pos, n, &used, &nb, &more))) DIE("gc_stream_eval_blocks at byte %zu: %d", pos, st);
                pos += used, done += nb;
                if (used == 0) {
                    if (!more || n == ftotal - pos) DIE("gc_stream_eval_blocks makes no progress at byte %zu (more %d)", pos, more);
                    win *= 2; /* a block larger than the piece */
                } else {
                    win = chunk;
                }
            }
            if (done != nsteps) DIE("gc_stream_eval_blocks evaluated %llu of %u blocks", (unsigned long long)done, nsteps);
            gc_label probe2 = {0, 0};
            if (gc_stream_eval_get_wire(e, step[nsteps - 1].out[0], &probe2)) DIE("gc_stream_eval_get_wire");
            eval_blocks_s = now_s() - t0;
            if (probe2.d0 != probe.d0 || probe2.d1 != probe.d1) DIE("gc_stream_eval_blocks: another label than block by block");
            gc_stream_eval_free(e);
        }
        /* ... and from a PINNED read buffer (gc_host_alloc: the DMA reads it in place) in pieces of 32 MiB: whole read buffers go
         * to the GPU, which recognises the blocks itself (mpc_amd/csrc/stream_eval_dev.cpp) */
        uint8_t *pin = (uint8_t *)gc_host_alloc(ftotal ? ftotal : 1);
        if (!pin) fprintf(stderr, "stream_driver: gc_host_alloc(%zu) refused [%s]: the pinned read-buffer pass is skipped\n", ftotal, gc_last_error());
        if (pin) {
            memcpy(pin, framed, ftotal);
            free(framed), framed = NULL;
            const size_t big = (size_t)32 << 20;
            /* (GC_DRIVER_PINNED_PASSES: a stress knob — how often the pass over the pinned read buffers is repeated) */
            const int npin = getenv("GC_DRIVER_PINNED_PASSES") ? atoi(getenv("GC_DRIVER_PINNED_PASSES")) : 2;
            for (int pass = 0; pass < npin; pass++) {
                gc_stream_eval *e = gc_stream_eval_create(ctx, key, keylen, &st);
                if (!e) DIE("gc_stream_eval_create: %d", st);
                for (uint32_t i = 0; i < nprim; i++)
                    if (gc_stream_eval_set_wire(e, prim[i], &in0[i])) DIE("gc_stream_eval_set_wire");
                const double t0 = now_s();
                size_t pos = 0, win = big;
                uint64_t done = 0;
                while (pos < ftotal) {
                    const size_t n = ftotal - pos < win ? ftotal - pos : win;
                    size_t used = 0;
                    uint32_t nb = 0;
                    int more = 0;
                    if ((st = gc_stream_eval_blocks(e, pin + pos, n, &used, &nb, &more))) DIE("gc_stream_eval_blocks (pinned) at byte %zu: %d", pos, st);
                    pos += used, done += nb;
                    if (used == 0) {
                        if (!more || n == ftotal - pos) DIE("gc_stream_eval_blocks (pinned) makes no progress at byte %zu", pos);
                        win *= 2;
                    } else {
                        win = big;
                    }
                }
                if (done != nsteps) DIE("gc_stream_eval_blocks (pinned) evaluated %llu of %u blocks", (unsigned long long)done, nsteps);
                gc_label probe3 = {0, 0};
                if (gc_stream_eval_get_wire(e, step[nsteps - 1].out[0], &probe3)) DIE("gc_stream_eval_get_wire");
                eval_blocks_pinned_s = now_s() - t0;
                if (probe3.d0 != probe.d0 || probe3.d1 != probe.d1) DIE("gc_stream_eval_blocks (pinned): another label than block by block");
                gc_stream_eval_free(e);
            }
            gc_host_free(pin);
        }
        free(framed);
    }
    int coop_state = 0;
    uint64_t coop_timeouts = 0;
    (void)gc_ctx_coop_stats(ctx, &coop_state, &coop_timeouts); /* 1: big steps ran as cooperative launches; -1: level launches */
    printf("{\"native\": true, \"steps\": %u, \"window\": %u, \"garble_s\": %.6f, \"garble_view_s\": %.6f, \"garble_async_s\": %.6f, \"eval_s\": %.6f, \"eval_steady_s\": %.6f, "
           "\"eval_blocks_s\": %.6f, \"eval_blocks_pinned_s\": %.6f, \"eval_blocks_chunk\": %zu, "
           "\"eval_steady_steps\": %u, \"bytes\": %zu, \"sha256\": \"%s\", "
           "\"eval_blocks_parsed\": %llu, \"eval_blocks_matched\": %llu, \"last_out_d0\": \"%016llx\", "
           "\"garble_t0\": %.6f, \"garble_t1\": %.6f, \"eval_t0\": %.6f, \"eval_t1\": %.6f, "
           "\"coop_state\": %d, \"coop_timeouts\": %llu, \"passes\": \"copying x2, deferred copies, views x2, eval per block x2, "
           "eval_blocks 1 MiB x2, eval_blocks pinned 32 MiB x2\"}\n",
           nsteps, window, garble_s, garble_view_s, garble_async_s, eval_s, eval_steady_s, eval_blocks_s, eval_blocks_pinned_s, chunk, eval_steady_steps, total, hex, (unsigned long long)parsed, (unsigned long long)matched,
           (unsigned long long)probe.d0, g_t0, g_t1, e_t0, e_t1, coop_state, (unsigned long long)coop_timeouts);
    gc_ctx_destroy(ctx);
    return 0;
}
