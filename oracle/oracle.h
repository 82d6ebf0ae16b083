/*
 * oracle.h — CPU restatement (plain C) of the markkurossi/mpc hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the CPU baseline, never as the thing shipped.
 *
 * The reference is Go and cannot be built here (no Go toolchain), so this is a
 * restatement, each function citing the reference file:line it follows.  The
 * AES / CTR arithmetic lives in the Go standard library (crypto/aes,
 * crypto/cipher, go 1.25.0 per go.mod:3), which is not under /root/reference; it
 * is restated from FIPS-197 / SP 800-38A.
 *
 * Pinning (tests/test_oracle_*.py): FIPS-197 App. C vectors, SP 800-38A F.5 CTR
 * vectors, OpenSSL cross-check (when libcrypto is loadable), the 8 MITCCRH
 * vectors of ot/mitccrh_test.go:23-30, the label arithmetic values of
 * ot/label_test.go:40-92, mul128 identities of ot/mul128_test.go, the decoded
 * digest of sha2pc/sha2pc_test.go:124 and the slab size of sha2pc/params.go:26.
 * Garbled-table BYTES are pinned by the reference's own transcript hashes
 * (sha2pc/sha2pc_test.go:121-123 and :413-415: SHA-256 of the encoded round 3 —
 * key, all 42 914 table labels of sha256xor.mpclc, input labels, output wires,
 * OT ciphertexts — of two deterministic runs): tests/go_transcript.py replays
 * those tests without Go (math/rand's seed table recomputed from its
 * definition, crypto/rand.Int, P-256, the sha2pc encodings) with orc_garble as
 * Circuit.Garble, and both constants come out (tests/test_go_transcript.py);
 * tests/test_gpu_go_transcript.py does the same with the HIP engine.
 */
#ifndef GC_ORACLE_H
#define GC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ot.Label — ot/label.go:28-31.  Memory layout identical to Go's struct. */
typedef struct { uint64_t d0, d1; } orc_label;
/* ot.Wire — ot/label.go:18-21 */
typedef struct { orc_label l0, l1; } orc_wire;
/* circuit.Gate — circuit/circuit.go:260-266 (20 bytes, circuit_test.go:14-19) */
typedef struct {
    uint32_t in0, in1, out;
    uint8_t op;
    uint8_t pad_[3];
    uint32_t level;
} orc_gate;

/* circuit.Operation — circuit/circuit.go:25-34 */
enum { ORC_XOR = 0, ORC_XNOR = 1, ORC_AND = 2, ORC_OR = 3, ORC_INV = 4 };

/* error codes (negative) */
enum {
    ORC_OK = 0,
    ORC_E_KEYSIZE = -1,   /* aes.NewCipher: invalid key size */
    ORC_E_RAND = -2,      /* rand reader exhausted */
    ORC_E_GATE = -3,      /* invalid gate type */
    ORC_E_ROWS = -4,      /* corrupted circuit: row length / index */
    ORC_E_ARG = -5,
};

/* ---- AES (FIPS-197); restates Go crypto/aes Block.Encrypt ------------- */
typedef struct {
    uint8_t rk[15 * 16]; /* round keys, byte order as in FIPS-197 w[] */
    int rounds;          /* 10 / 12 / 14 */
} orc_aes;

int  orc_aes_init(orc_aes *a, const uint8_t *key, size_t keylen);
void orc_aes_encrypt(const orc_aes *a, const uint8_t in[16], uint8_t out[16]);
/* force the portable byte-wise path (1) or allow AES-NI when the CPU has it (0) */
void orc_aes_force_portable(int on);
int  orc_aes_using_aesni(void);

/* ---- labels: ot/label.go ---------------------------------------------- */
void orc_label_mul2(orc_label *l);
void orc_label_mul4(orc_label *l);
void orc_label_set_s(orc_label *l, int set);
int  orc_label_s(const orc_label *l);
unsigned orc_label_bit(const orc_label *l, int i);
void orc_label_get_data(const orc_label *l, uint8_t out[16]);
void orc_label_set_data(orc_label *l, const uint8_t in[16]);

/* ---- garbling primitives: circuit/garble.go:40-143 --------------------- */
orc_label orc_encrypt_half(const orc_aes *alg, orc_label x, uint32_t i);
orc_label orc_encrypt(const orc_aes *alg, orc_label a, orc_label b, orc_label c, uint32_t t);
orc_label orc_decrypt(const orc_aes *alg, orc_label a, orc_label b, uint32_t t, orc_label c);

/* ---- Circuit.Garble: circuit/garble.go:248-308 ------------------------- */
/* rnd is the byte stream the Go io.Reader would deliver: R (16 B) then one
 * 16-byte L0 per input wire (R0 order).  wires[nwires] and
 * slab[slab_cap] are filled; gate_off[ngates+1] (optional) receives each gate's
 * first slab row (gate i owns rows [gate_off[i], gate_off[i+1])).
 * Returns the number of slab rows written or a negative error. */
long orc_garble(const orc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                const uint8_t *key, size_t keylen, const uint8_t *rnd, size_t rndlen,
                orc_label *r_out, orc_wire *wires, orc_label *slab, size_t slab_cap,
                uint32_t *gate_off);

/* ---- Circuit.Eval: circuit/eval.go:17-115 ------------------------------ */
/* wires[nwires] holds the active labels of the input wires at [0,ninputs) and is
 * written in place.  slab/slab_rows is the dense table in gate order. */
int orc_eval(const orc_gate *gates, uint32_t ngates, uint32_t nwires,
             const uint8_t *key, size_t keylen, orc_label *wires,
             const orc_label *slab, size_t slab_rows);

/* ---- Circuit.Compute: circuit/computer.go:15-91 (plaintext truth) ------ */
int orc_compute(const orc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                const uint8_t *in_bits, uint8_t *wire_bits);

/* ---- AssignLevels(TargetYao): circuit/circuit.go:206-254 --------------- */
/* fills gates[i].level; returns number of levels (Stats[NumLevels]); *max_width */
uint32_t orc_assign_levels(orc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t *max_width);

/* ---- AES-128-CTR PRG: ot/iknp.go:622-645 ------------------------------- */
typedef struct {
    orc_aes aes;
    uint8_t ctr[16];  /* big-endian 128-bit counter (Go cipher.NewCTR, zero IV) */
    uint8_t ks[16];   /* current keystream block */
    int used;         /* bytes of ks already consumed (16 = none buffered) */
} orc_prg;
void orc_prg_init(orc_prg *p, orc_label key);
void orc_prg_bytes(orc_prg *p, uint8_t *buf, size_t n); /* prg(): buf = keystream */
void orc_prg_labels(orc_prg *p, orc_label *labels, size_t n);

/* ---- IKNP: ot/iknp.go ------------------------------------------------- */
#define ORC_IKNP_K 128
#define ORC_IKNP_CHUNK 8192
typedef struct { orc_prg g0[ORC_IKNP_K]; orc_prg g1[ORC_IKNP_K]; } orc_iknp_receiver;
typedef struct { orc_prg g0[ORC_IKNP_K]; orc_label delta; } orc_iknp_sender;

void orc_iknp_receiver_init(orc_iknp_receiver *r, const orc_wire base[ORC_IKNP_K]);       /* iknp.go:347-356 */
void orc_iknp_sender_init(orc_iknp_sender *s, orc_label delta, const orc_label k0[ORC_IKNP_K]); /* iknp.go:117-122 */
/* receive(): iknp.go:468-511.  b[n] are bools.  u_out receives the chunks the
 * receiver would SendData, concatenated (each chunk = byteRows*128 bytes).
 * Returns total bytes written to u_out. */
size_t orc_iknp_receive(orc_iknp_receiver *r, const uint8_t *b, size_t n, uint8_t *u_out, orc_label *result);
/* send(): iknp.go:197-226, consuming the concatenated chunks in u_in.
 * chunk boundaries are recomputed the way the receiver produced them. */
size_t orc_iknp_send(orc_iknp_sender *s, const uint8_t *u_in, size_t n, orc_label *result);
/* bit-COT: (*IKNPSender).SendBits iknp.go:259-310 / (*IKNPReceiver).ReceiveBits iknp.go:554-620.
 * choices / result are packed little-endian bit vectors (bit i = word i/64, position i%64).
 * Only column 0 of the matrix survives: r_i = s_i ^ (b_i & Delta.Bit(0)). */
size_t orc_iknp_receive_bits(orc_iknp_receiver *r, const uint64_t *choices, size_t n, uint8_t *u_out, uint64_t *result);
size_t orc_iknp_send_bits(orc_iknp_sender *s, const uint8_t *u_in, size_t n, uint64_t *result);
/* createLabels(): iknp.go:647-683 */
void orc_create_labels(orc_label *l, size_t nl, const uint8_t *buf, int w);

/* ---- MITCCRH: ot/mitccrh.go:50-128 ------------------------------------ */
typedef struct {
    int batch_size;
    orc_label start;
    uint64_t gid;
    orc_aes ciphers[8];
    int key_used;
} orc_mitccrh;
void orc_mitccrh_init(orc_mitccrh *m, orc_label seed, int batch_size /* <= 8 */);
void orc_mitccrh_hash(orc_mitccrh *m, orc_label *blks, int k, int h);

/* COT.Send / COT.Receive post-processing: ot/cot.go:136-235 (wire bytes) */
/* out[2*n] = the labels the sender would SendLabel, in order */
void orc_cot_send_pads(orc_label seed, orc_label delta, const orc_label *data, const orc_wire *wires, size_t n, orc_label *out);
/* result[n] in: IKNP receive output; out: chosen labels */
void orc_cot_receive_unpad(orc_label seed, const uint8_t *flags, const orc_label *sent /*2n*/, orc_label *result, size_t n);
/* ROT pad loops: ot/rot.go:156-172 (sender: wires[j] = the two hashed pads) and :194-199 (receiver: in place) */
void orc_rot_send(orc_label seed, orc_label delta, const orc_label *data, size_t n, orc_wire *wires);
void orc_rot_receive(orc_label seed, orc_label *result, size_t n);

/* ---- GF(2^128): ot/mul128_generic.go, ot/gf128.go ---------------------- */
void orc_mul128(orc_label a, orc_label b, orc_label *lo, orc_label *hi);
void orc_mul128_ref(orc_label a, orc_label b, orc_label *lo, orc_label *hi); /* mul128_ref.go:9 */
void orc_inner_product(const orc_label *a, const orc_label *b, size_t n, orc_label *r1, orc_label *r2);

/* ---- KOS consistency check of the malicious IKNP variant: ot/iknp.go:138-194 (sender), :373-465 (receiver) ----
 * chi_i = i-th label of the AES-128-CTR stream keyed by seed2 (prgLabels, iknp.go:639-645): indices
 * 0..n-1 for the extension results, n..n+255 for the 256-label random choice vector.
 * receiver: t = XOR_i chi_i * result_i (256-bit, no reduction, gf128.go:14-27), x = XOR_{b_i} chi_i */
void orc_kos_receiver_tags(orc_label seed2, const orc_label *result, const uint8_t *b, size_t n,
                           const orc_label *choice_vec, const uint8_t *bcv /* 256 */, orc_label *x,
                           orc_label *t0, orc_label *t1);
/* sender: q = XOR_i chi_i * result_i ^ x * Delta; returns 1 iff (q0,q1) == (t0,t1)  ("OT extension check failed") */
int orc_kos_sender_check(orc_label seed2, const orc_label *result, size_t n, const orc_label *choice_vec,
                         orc_label delta, orc_label x, orc_label t0, orc_label t1);

/* ---- garbled-table wire format of the 2-party driver ------------------------------------------- */
/* circuit/garbler.go:69-82: SendUint32(#gates), then per gate SendUint32(len) + SendLabel per row
 * (big-endian u32, labels as BE(D0)||BE(D1)).  Returns bytes written (4 + 4*ngates + 16*rows). */
size_t orc_tables_serialize(const orc_gate *gates, uint32_t ngates, const orc_label *slab, uint8_t *out);
/* circuit/evaluator.go:40-66: parse the same bytes back into a dense slab; negative on a gate count
 * mismatch ("wrong number of gates") or a row count that does not fit the gate type */
long orc_tables_parse(const orc_gate *gates, uint32_t ngates, const uint8_t *in, size_t len, orc_label *slab);

/* ---- Streaming garbler / evaluator: circuit/stream_garble.go, circuit/stream_evaluator.go ---- */
typedef struct orc_stream orc_stream;
/* NewStreaming (stream_garble.go:41-75): rnd = R (16 B) then one L0 per entry of inputs[] */
orc_stream *orc_stream_new(const uint8_t *key, size_t keylen, const uint8_t *rnd, size_t rndlen,
                           const uint32_t *inputs, uint32_t ninputs, int *status);
void orc_stream_free(orc_stream *);
/* Streaming.GetInput (stream_garble.go:117-119) */
int orc_stream_get(orc_stream *, uint32_t w, orc_wire *out);
/* Streaming.Garble (stream_garble.go:161-192): appends the serialised gates (:391-446) to buf;
 * returns bytes written or a negative error (ORC_E_ARG if cap is too small) */
long orc_stream_garble(orc_stream *, const orc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                       uint32_t nin, const uint32_t *out, uint32_t nout, uint8_t *buf, size_t cap);
/* evaluator: StreamEval state + the per-gate loop of stream_evaluator.go:271-432 over one circuit's bytes */
typedef struct orc_stream_eval orc_stream_eval;
orc_stream_eval *orc_stream_eval_new(const uint8_t *key, size_t keylen, int *status);
void orc_stream_eval_free(orc_stream_eval *);
int orc_stream_eval_set(orc_stream_eval *, uint32_t w, orc_label l);
int orc_stream_eval_get(orc_stream_eval *, uint32_t w, orc_label *l);
/* returns bytes consumed or negative error */
long orc_stream_eval_circuit(orc_stream_eval *, uint32_t ngates, uint32_t ntmp, uint32_t nwires, const uint8_t *buf,
                             size_t len);

/* ---- CPU baseline helper: garble+eval `reps` instances on `threads` ---- */
/* returns elapsed seconds; checks every evaluated output label against the
 * garbler's wire labels (returns negative on mismatch). */
double orc_bench_garble_eval(const orc_gate *gates, uint32_t ngates, uint32_t nwires,
                             uint32_t ninputs, uint32_t noutputs, const uint8_t *key, size_t keylen,
                             uint32_t reps, int threads);

#ifdef __cplusplus
}
#endif
#endif
