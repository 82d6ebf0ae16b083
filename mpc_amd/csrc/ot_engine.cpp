// ot_engine.cpp — C ABI for the IKNP OT extension, MITCCRH and the COT pad loops.
#include <algorithm>
#include <cstring>
#include <new>

#include "engine.h"

using namespace gc;

struct gc_iknp {
    gc_ctx *ctx = nullptr;
    bool receiver = false;
    uint32_t *d_rk0 = nullptr, *d_rk1 = nullptr;  // [128][44]
    uint64_t pos = 0;                             // bytes drawn so far from every column stream
    uint4 delta{};
    // device-resident API (gc_iknp_*_dev): workspace kept across calls, events around the kernels
    void *d_ws = nullptr;
    size_t ws_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
};

namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

inline uint4 to_u4(const gc_label &l) {
    return make_uint4((uint32_t)l.d0, (uint32_t)(l.d0 >> 32), (uint32_t)l.d1, (uint32_t)(l.d1 >> 32));
}

// newPrg (iknp.go:622-630): AES-128 key = BE(label)
void expand_label_key(const gc_label &l, uint32_t *out44) {
    uint8_t kb[16];
    for (int i = 0; i < 8; i++) {
        kb[i] = (uint8_t)(l.d0 >> (56 - 8 * i));
        kb[8 + i] = (uint8_t)(l.d1 >> (56 - 8 * i));
    }
    AesKey k;
    aes_expand_key(kb, 16, &k);
    std::memcpy(out44, k.w, 44 * sizeof(uint32_t));
}

int upload_keys(gc_ctx *ctx, const std::vector<uint32_t> &host, uint32_t **dptr) {
    GC_HIP(hipMalloc((void **)dptr, host.size() * sizeof(uint32_t)));
    GC_HIP(hipMemcpy(*dptr, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    (void)ctx;
    return GC_OK;
}

}  // namespace

extern "C" {

gc_iknp *gc_iknp_receiver_create(gc_ctx *ctx, const gc_wire *base, int *status) try {
    int rc = (ctx && base) ? GC_OK : GC_E_ARG;
    gc_iknp *k = nullptr;
    if (rc == GC_OK && !(k = new (std::nothrow) gc_iknp)) rc = GC_E_NOMEM;
    if (rc == GC_OK) {
        k->ctx = ctx;
        k->receiver = true;
        std::vector<uint32_t> r0(128 * 44), r1(128 * 44);
        for (int i = 0; i < 128; i++) {
            expand_label_key(base[i].l0, &r0[44 * i]);
            expand_label_key(base[i].l1, &r1[44 * i]);
        }
        if (hipSetDevice(ctx->device) != hipSuccess) rc = GC_E_HIP;
        if (rc == GC_OK) rc = upload_keys(ctx, r0, &k->d_rk0);
        if (rc == GC_OK) rc = upload_keys(ctx, r1, &k->d_rk1);
    }
    if (rc != GC_OK && k) {
        gc_iknp_free(k);
        k = nullptr;
    }
    if (status) *status = rc;
    return k;
} catch (...) {
    const int rc__ = gc::on_exception();
    if (status) *status = rc__;
    return nullptr;
}

gc_iknp *gc_iknp_sender_create(gc_ctx *ctx, const gc_label *delta, const gc_label *k0, int *status) try {
    int rc = (ctx && delta && k0) ? GC_OK : GC_E_ARG;
    gc_iknp *k = nullptr;
    if (rc == GC_OK && !(k = new (std::nothrow) gc_iknp)) rc = GC_E_NOMEM;
    if (rc == GC_OK) {
        k->ctx = ctx;
        k->receiver = false;
        k->delta = to_u4(*delta);
        std::vector<uint32_t> r0(128 * 44);
        for (int i = 0; i < 128; i++) expand_label_key(k0[i], &r0[44 * i]);
        if (hipSetDevice(ctx->device) != hipSuccess) rc = GC_E_HIP;
        if (rc == GC_OK) rc = upload_keys(ctx, r0, &k->d_rk0);
    }
    if (rc != GC_OK && k) {
        gc_iknp_free(k);
        k = nullptr;
    }
    if (status) *status = rc;
    return k;
} catch (...) {
    const int rc__ = gc::on_exception();
    if (status) *status = rc__;
    return nullptr;
}

void gc_iknp_free(gc_iknp *k) {
    if (!k) return;
    if (k->ctx) (void)hipSetDevice(k->ctx->device);
    if (k->d_rk0) (void)hipFree(k->d_rk0);
    if (k->d_rk1) (void)hipFree(k->d_rk1);
    if (k->d_ws) (void)hipFree(k->d_ws);
    if (k->ev0) (void)hipEventDestroy(k->ev0);
    if (k->ev1) (void)hipEventDestroy(k->ev1);
    delete k;
}

size_t gc_iknp_u_bytes(size_t n) {
    // full chunks: 512 rows = 64 byte-rows x 128 columns; the last chunk has ceil(rows/8) byte-rows
    size_t full = n / 512, rem = n % 512;
    return full * 8192 + ((rem + 7) / 8) * 128;
}

static size_t stream_advance(size_t n) { return (n / 512) * 64 + ((n % 512) + 7) / 8; }

// receive() with the choice bits already packed LSB-first, 64 bytes per chunk (bbuf[chunks*64])
static int iknp_receive_packed(gc_iknp *k, const std::vector<uint8_t> &bbuf, size_t n, uint8_t *u_out,
                               gc_label *labels_out) {
    gc_ctx *ctx = k->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GC_HIP(hipSetDevice(ctx->device));
    const size_t chunks = (n + 511) / 512, ub = gc_iknp_u_bytes(n);
    DevBuf d_bits, d_u, d_lab;
    GC_HIP(d_bits.alloc(chunks * 64 + 16));
    GC_HIP(d_u.alloc(chunks * 8192));
    GC_HIP(d_lab.alloc(n * sizeof(uint4)));
    hipStream_t s = ctx->stream;
    GC_HIP(hipMemsetAsync(d_bits.p, 0, chunks * 64 + 16, s));
    GC_HIP(hipMemcpyAsync(d_bits.p, bbuf.data(), std::min(bbuf.size(), chunks * 64), hipMemcpyHostToDevice, s));
    GC_HIP(launch_iknp_fused(true, k->d_rk0, k->d_rk1, k->pos, n, (const uint8_t *)d_bits.p, nullptr, k->delta,
                             (uint8_t *)d_u.p, (uint4 *)d_lab.p, ctx->d_te0, s));
    GC_HIP(hipMemcpyAsync(u_out, d_u.p, ub, hipMemcpyDeviceToHost, s));
    GC_HIP(hipMemcpyAsync(labels_out, d_lab.p, n * sizeof(uint4), hipMemcpyDeviceToHost, s));
    GC_HIP(hipStreamSynchronize(s));
    k->pos += stream_advance(n);
    return GC_OK;
}

int gc_iknp_receive(gc_iknp *k, const uint8_t *choice, size_t n, uint8_t *u_out, gc_label *labels_out) try {
    if (!k || !k->receiver || (n && (!choice || !u_out || !labels_out))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    std::vector<uint8_t> bbuf(((n + 511) / 512) * 64, 0);  // iknp.go:472-477
    for (size_t i = 0; i < n; i++)
        if (choice[i]) bbuf[i / 8] |= (uint8_t)(1u << (i % 8));
    return iknp_receive_packed(k, bbuf, n, u_out, labels_out);
} catch (...) {
    return gc::on_exception();
}

// (*IKNPReceiver).ReceiveBits (iknp.go:554-620): same matrix, result = bit 0 of every label.  The reference
// folds the choice vector in as whole little-endian 64-bit words only (words = byteRows/8, :583-597): the
// choice bits of a trailing partial word do NOT enter u — reproduced here bit for bit.
int gc_iknp_receive_bits(gc_iknp *k, const uint64_t *choices, size_t n, uint8_t *u_out, uint64_t *result) try {
    if (!k || !k->receiver || (n && (!choices || !u_out || !result))) return GC_E_ARG;
    for (size_t i = 0; i < (n + 63) / 64; i++) result[i] = 0;
    if (n == 0) return GC_OK;
    const size_t chunks = (n + 511) / 512;
    std::vector<uint8_t> bbuf(chunks * 64, 0);
    for (size_t c = 0; c < chunks; c++) {
        const size_t ofs = c * 512, rows = std::min<size_t>(512, n - ofs), words = ((rows + 7) / 8) / 8;
        for (size_t w = 0; w < words; w++)
            for (int b = 0; b < 8; b++) bbuf[c * 64 + w * 8 + b] = (uint8_t)(choices[ofs / 64 + w] >> (8 * b));
    }
    std::vector<gc_label> labels(n);
    int rc = iknp_receive_packed(k, bbuf, n, u_out, labels.data());
    if (rc != GC_OK) return rc;
    for (size_t i = 0; i < n; i++)
        if (labels[i].d0 & 1) result[i / 64] |= (uint64_t)1 << (i % 64);  // labelsBuf[row].Bit(0) (:609-614)
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_iknp_send(gc_iknp *k, const uint8_t *u_in, size_t u_len, size_t n, gc_label *labels_out) try {
    if (!k || k->receiver || (n && (!u_in || !labels_out))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    if (u_len != gc_iknp_u_bytes(n)) return GC_E_ARG;  // "invalid chunk size" (iknp.go:207-209)
    gc_ctx *ctx = k->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GC_HIP(hipSetDevice(ctx->device));
    const size_t chunks = (n + 511) / 512;
    DevBuf d_u, d_lab;
    GC_HIP(d_u.alloc(chunks * 8192));
    GC_HIP(d_lab.alloc(n * sizeof(uint4)));
    hipStream_t s = ctx->stream;
    GC_HIP(hipMemcpyAsync(d_u.p, u_in, u_len, hipMemcpyHostToDevice, s));
    GC_HIP(launch_iknp_fused(false, k->d_rk0, nullptr, k->pos, n, nullptr, (const uint8_t *)d_u.p, k->delta, nullptr,
                             (uint4 *)d_lab.p, ctx->d_te0, s));
    GC_HIP(hipMemcpyAsync(labels_out, d_lab.p, n * sizeof(uint4), hipMemcpyDeviceToHost, s));
    GC_HIP(hipStreamSynchronize(s));
    k->pos += stream_advance(n);
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

// ---- device-resident entry points: no host staging, no allocation after the first call, asynchronous on the
// ctx stream (gc_ctx_sync to wait).  Same kernels, same bytes as gc_iknp_receive / gc_iknp_send.
static int iknp_ws(gc_iknp *k, size_t bytes) {
    if (!k->ev0) {
        GC_HIP(hipEventCreate(&k->ev0));
        GC_HIP(hipEventCreate(&k->ev1));
    }
    if (bytes <= k->ws_bytes) return GC_OK;
    GC_HIP(hipStreamSynchronize(k->ctx->stream));
    if (k->d_ws) (void)hipFree(k->d_ws);
    k->d_ws = nullptr;
    k->ws_bytes = 0;
    GC_HIP(hipMalloc(&k->d_ws, bytes));
    k->ws_bytes = bytes;
    return GC_OK;
}

int gc_iknp_receive_dev(gc_iknp *k, const void *d_choice_packed, size_t n, void *d_u_out, void *d_labels_out) {
    if (!k || !k->receiver || (n && (!d_choice_packed || !d_u_out || !d_labels_out))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    gc_ctx *ctx = k->ctx;
    // the stream position is a kernel argument and advances with every call: a replayed capture would reuse it
    if (ctx->capturing) return GC_E_ARG;
    GC_HIP(hipSetDevice(ctx->device));
    int rc = iknp_ws(k, 0);
    if (rc != GC_OK) return rc;
    hipStream_t s = ctx->stream;
    GC_HIP(hipEventRecord(k->ev0, s));
    GC_HIP(launch_iknp_fused(true, k->d_rk0, k->d_rk1, k->pos, n, (const uint8_t *)d_choice_packed, nullptr, k->delta,
                             (uint8_t *)d_u_out, (uint4 *)d_labels_out, ctx->d_te0, s));
    GC_HIP(hipEventRecord(k->ev1, s));
    k->timed = true;
    k->pos += stream_advance(n);
    return GC_OK;
}

int gc_iknp_send_dev(gc_iknp *k, const void *d_u_in, size_t n, void *d_labels_out) {
    if (!k || k->receiver || (n && (!d_u_in || !d_labels_out))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    gc_ctx *ctx = k->ctx;
    // the stream position is a kernel argument and advances with every call: a replayed capture would reuse it
    if (ctx->capturing) return GC_E_ARG;
    GC_HIP(hipSetDevice(ctx->device));
    int rc = iknp_ws(k, 0);
    if (rc != GC_OK) return rc;
    hipStream_t s = ctx->stream;
    GC_HIP(hipEventRecord(k->ev0, s));
    GC_HIP(launch_iknp_fused(false, k->d_rk0, nullptr, k->pos, n, nullptr, (const uint8_t *)d_u_in, k->delta, nullptr,
                             (uint4 *)d_labels_out, ctx->d_te0, s));
    GC_HIP(hipEventRecord(k->ev1, s));
    k->timed = true;
    k->pos += stream_advance(n);
    return GC_OK;
}

// bit-COT with everything in HBM (additive): choices / result are packed little-endian u64 DEVICE
// vectors of (n + 63) / 64 words, d_u the u-matrix as in gc_iknp_*_dev.  The labels go through the handle's
// workspace and never leave the device.
int gc_iknp_receive_bits_dev(gc_iknp *k, const void *d_choices, size_t n, void *d_u_out, void *d_result) {
    if (!k || !k->receiver || (n && (!d_choices || !d_u_out || !d_result))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    gc_ctx *ctx = k->ctx;
    if (ctx->capturing) return GC_E_ARG;
    GC_HIP(hipSetDevice(ctx->device));
    const size_t chunks = (n + 511) / 512;
    int rc = iknp_ws(k, n * sizeof(uint4) + chunks * 64);
    if (rc != GC_OK) return rc;
    uint4 *lab = (uint4 *)k->d_ws;
    uint64_t *bbuf = (uint64_t *)((uint8_t *)k->d_ws + n * sizeof(uint4));
    hipStream_t s = ctx->stream;
    GC_HIP(hipEventRecord(k->ev0, s));
    launch_fold_choice_words((const uint64_t *)d_choices, n, bbuf, s);
    GC_HIP(launch_iknp_fused(true, k->d_rk0, k->d_rk1, k->pos, n, (const uint8_t *)bbuf, nullptr, k->delta,
                             (uint8_t *)d_u_out, lab, ctx->d_te0, s));
    launch_pack_label_bit0(lab, n, (uint64_t *)d_result, s);
    GC_HIP(hipGetLastError());
    GC_HIP(hipEventRecord(k->ev1, s));
    k->timed = true;
    k->pos += stream_advance(n);
    return GC_OK;
}

int gc_iknp_send_bits_dev(gc_iknp *k, const void *d_u_in, size_t n, void *d_result) {
    if (!k || k->receiver || (n && (!d_u_in || !d_result))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    gc_ctx *ctx = k->ctx;
    if (ctx->capturing) return GC_E_ARG;
    GC_HIP(hipSetDevice(ctx->device));
    int rc = iknp_ws(k, n * sizeof(uint4));
    if (rc != GC_OK) return rc;
    uint4 *lab = (uint4 *)k->d_ws;
    hipStream_t s = ctx->stream;
    GC_HIP(hipEventRecord(k->ev0, s));
    GC_HIP(launch_iknp_fused(false, k->d_rk0, nullptr, k->pos, n, nullptr, (const uint8_t *)d_u_in, k->delta, nullptr, lab,
                             ctx->d_te0, s));
    launch_pack_label_bit0(lab, n, (uint64_t *)d_result, s);
    GC_HIP(hipGetLastError());
    GC_HIP(hipEventRecord(k->ev1, s));
    k->timed = true;
    k->pos += stream_advance(n);
    return GC_OK;
}

float gc_iknp_last_ms(gc_iknp *k) {
    if (!k || !k->timed) return -1.0f;
    if (hipSetDevice(k->ctx->device) != hipSuccess) return -1.0f;
    if (hipEventSynchronize(k->ev1) != hipSuccess) return -1.0f;
    float ms = 0;
    if (hipEventElapsedTime(&ms, k->ev0, k->ev1) != hipSuccess) return -1.0f;
    return ms;
}

// ---- KOS consistency check --------------------------------------------------------------------------------

static void host_clmul128(const gc_label &a, const gc_label &b, uint64_t out[4]) {  // mul128_generic.go
    out[0] = out[1] = out[2] = out[3] = 0;
    const uint64_t aw[2] = {a.d0, a.d1}, bw[2] = {b.d0, b.d1};
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
            uint64_t lo = 0, hi = 0;
            for (int k = 0; k < 64; k++)
                if ((bw[j] >> k) & 1) {
                    lo ^= aw[i] << k;
                    if (k) hi ^= aw[i] >> (64 - k);
                }
            out[i + j] ^= lo;
            out[i + j + 1] ^= hi;
        }
}

// acc[6] <- sums over (result, b) then (choice_vec, bcv); shared by both roles
// chi-PRG + inner products over labels that are ALREADY on the device (d_res [n], d_bits [n] or null) plus the 256-label
// choice vector from the host: two launches into one accumulator — chi labels are consecutive in the stream, results
// first (counter 0..n-1), then the choice vector (n..n+255) (iknp.go:159-174)
static int kos_sums_dev(gc_ctx *ctx, const gc_label *seed2, const uint4 *d_res, const uint8_t *d_bits, size_t n,
                        const gc_label *choice_vec, const uint8_t *bcv, uint64_t acc[6]) {
    GC_HIP(hipSetDevice(ctx->device));
    uint32_t rk[44];
    expand_label_key(*seed2, rk);  // newPrg(seed2) (iknp.go:150, 411)
    DevBuf d_rk, d_cv, d_cb, d_zero, d_acc;
    GC_HIP(d_rk.alloc(sizeof rk));
    GC_HIP(d_cv.alloc(256 * sizeof(uint4)));
    GC_HIP(d_cb.alloc(256));
    GC_HIP(d_acc.alloc(6 * sizeof(uint64_t)));
    hipStream_t s = ctx->stream;
    GC_HIP(hipMemcpyAsync(d_rk.p, rk, sizeof rk, hipMemcpyHostToDevice, s));
    GC_HIP(hipMemcpyAsync(d_cv.p, choice_vec, 256 * sizeof(uint4), hipMemcpyHostToDevice, s));
    GC_HIP(hipMemsetAsync(d_cb.p, 0, 256, s));
    if (bcv) GC_HIP(hipMemcpyAsync(d_cb.p, bcv, 256, hipMemcpyHostToDevice, s));
    if (n && !d_bits) {  // the sender has no choice bits: x is not used, feed zeros
        GC_HIP(d_zero.alloc(n));
        GC_HIP(hipMemsetAsync(d_zero.p, 0, n, s));
        d_bits = (const uint8_t *)d_zero.p;
    }
    GC_HIP(hipMemsetAsync(d_acc.p, 0, 6 * sizeof(uint64_t), s));
    launch_kos_accumulate((const uint32_t *)d_rk.p, 0, d_res, d_bits, n, (unsigned long long *)d_acc.p, ctx->d_te0, s);
    launch_kos_accumulate((const uint32_t *)d_rk.p, n, (const uint4 *)d_cv.p, (const uint8_t *)d_cb.p, 256,
                          (unsigned long long *)d_acc.p, ctx->d_te0, s);
    GC_HIP(hipGetLastError());
    GC_HIP(hipMemcpyAsync(acc, d_acc.p, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    GC_HIP(hipStreamSynchronize(s));
    return GC_OK;
}

// host-buffer form: stage the labels (and choice bits), then as above
static int kos_sums(gc_ctx *ctx, const gc_label *seed2, const gc_label *result, const uint8_t *b, size_t n,
                    const gc_label *choice_vec, const uint8_t *bcv, uint64_t acc[6]) {
    GC_HIP(hipSetDevice(ctx->device));
    DevBuf d_v, d_b;
    hipStream_t s = ctx->stream;
    if (n) {
        GC_HIP(d_v.alloc(n * sizeof(uint4)));
        GC_HIP(hipMemcpyAsync(d_v.p, result, n * sizeof(uint4), hipMemcpyHostToDevice, s));
        if (b) {
            GC_HIP(d_b.alloc(n));
            GC_HIP(hipMemcpyAsync(d_b.p, b, n, hipMemcpyHostToDevice, s));
        }
    }
    return kos_sums_dev(ctx, seed2, (const uint4 *)d_v.p, b ? (const uint8_t *)d_b.p : nullptr, n, choice_vec, bcv, acc);
}

static int kos_finish_sender(const uint64_t acc[6], const gc_label *delta, const gc_label *x, const gc_label *t0,
                             const gc_label *t1) {
    uint64_t r[4];
    host_clmul128(*x, *delta, r);  // mul128(x, s.Delta) (iknp.go:186)
    return (acc[0] ^ r[0]) == t0->d0 && (acc[1] ^ r[1]) == t0->d1 && (acc[2] ^ r[2]) == t1->d0 &&
           (acc[3] ^ r[3]) == t1->d1;
}

// device-resident forms: the labels gc_iknp_receive_dev / gc_iknp_send_dev left in HBM are checked where they are
int gc_kos_receiver_tags_dev(gc_ctx *ctx, const gc_label *seed2, const void *d_result, const void *d_b, size_t n,
                             const gc_label *choice_vec, const uint8_t *bcv, gc_label *x, gc_label *t0, gc_label *t1) try {
    if (!ctx || !seed2 || !choice_vec || !bcv || !x || !t0 || !t1 || (n && (!d_result || !d_b))) return GC_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    uint64_t acc[6];
    int rc = kos_sums_dev(ctx, seed2, (const uint4 *)d_result, (const uint8_t *)d_b, n, choice_vec, bcv, acc);
    if (rc != GC_OK) return rc;
    *t0 = gc_label{acc[0], acc[1]};
    *t1 = gc_label{acc[2], acc[3]};
    *x = gc_label{acc[4], acc[5]};
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_kos_sender_check_dev(gc_ctx *ctx, const gc_label *seed2, const void *d_result, size_t n,
                            const gc_label *choice_vec, const gc_label *delta, const gc_label *x, const gc_label *t0,
                            const gc_label *t1, int *ok) try {
    if (!ctx || !seed2 || !choice_vec || !delta || !x || !t0 || !t1 || !ok || (n && !d_result)) return GC_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    uint64_t acc[6];
    int rc = kos_sums_dev(ctx, seed2, (const uint4 *)d_result, nullptr, n, choice_vec, nullptr, acc);
    if (rc != GC_OK) return rc;
    *ok = kos_finish_sender(acc, delta, x, t0, t1);
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_kos_receiver_tags(gc_ctx *ctx, const gc_label *seed2, const gc_label *result, const uint8_t *b, size_t n,
                         const gc_label *choice_vec, const uint8_t *bcv, gc_label *x, gc_label *t0, gc_label *t1) try {
    if (!ctx || !seed2 || !choice_vec || !bcv || !x || !t0 || !t1 || (n && (!result || !b))) return GC_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    uint64_t acc[6];
    int rc = kos_sums(ctx, seed2, result, b, n, choice_vec, bcv, acc);
    if (rc != GC_OK) return rc;
    *t0 = gc_label{acc[0], acc[1]};
    *t1 = gc_label{acc[2], acc[3]};
    *x = gc_label{acc[4], acc[5]};
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_kos_sender_check(gc_ctx *ctx, const gc_label *seed2, const gc_label *result, size_t n,
                        const gc_label *choice_vec, const gc_label *delta, const gc_label *x, const gc_label *t0,
                        const gc_label *t1, int *ok) try {
    if (!ctx || !seed2 || !choice_vec || !delta || !x || !t0 || !t1 || !ok || (n && !result)) return GC_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    uint64_t acc[6];
    int rc = kos_sums(ctx, seed2, result, nullptr, n, choice_vec, nullptr, acc);
    if (rc != GC_OK) return rc;
    *ok = kos_finish_sender(acc, delta, x, t0, t1);
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

// (*IKNPSender).SendBits (iknp.go:259-310): column 0 of the q-matrix == bit 0 of every label of send()
int gc_iknp_send_bits(gc_iknp *k, const uint8_t *u_in, size_t u_len, size_t n, uint64_t *result) try {
    if (!k || k->receiver || (n && (!u_in || !result))) return GC_E_ARG;
    for (size_t i = 0; i < (n + 63) / 64; i++) result[i] = 0;
    if (n == 0) return GC_OK;
    std::vector<gc_label> labels(n);
    int rc = gc_iknp_send(k, u_in, u_len, n, labels.data());
    if (rc != GC_OK) return rc;
    for (size_t i = 0; i < n; i++)
        if (labels[i].d0 & 1) result[i / 64] |= (uint64_t)1 << (i % 64);
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_mitccrh_hash(gc_ctx *ctx, const gc_label *seed, uint64_t gid0, gc_label *blks, size_t n, uint32_t h) try {
    if (!ctx || !seed || (n && !blks)) return GC_E_ARG;
    if (n == 0 || h == 0) return GC_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GC_HIP(hipSetDevice(ctx->device));
    DevBuf d;
    const size_t bytes = n * h * sizeof(uint4);
    GC_HIP(d.alloc(bytes));
    GC_HIP(hipMemcpyAsync(d.p, blks, bytes, hipMemcpyHostToDevice, ctx->stream));
    launch_mitccrh(to_u4(*seed), gid0, (uint4 *)d.p, n, h, ctx->d_te0, ctx->stream);
    GC_HIP(hipGetLastError());
    GC_HIP(hipMemcpyAsync(blks, d.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_cot_send_pads(gc_ctx *ctx, const gc_label *seed, const gc_label *delta, const gc_label *data,
                     const gc_wire *wires, size_t n, gc_label *out) try {
    if (!ctx || !seed || !delta || (n && (!data || !wires || !out))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GC_HIP(hipSetDevice(ctx->device));
    DevBuf d_data, d_w, d_out;
    GC_HIP(d_data.alloc(n * 16));
    GC_HIP(d_w.alloc(n * 32));
    GC_HIP(d_out.alloc(n * 32));
    hipStream_t s = ctx->stream;
    GC_HIP(hipMemcpyAsync(d_data.p, data, n * 16, hipMemcpyHostToDevice, s));
    GC_HIP(hipMemcpyAsync(d_w.p, wires, n * 32, hipMemcpyHostToDevice, s));
    launch_cot_send(to_u4(*seed), to_u4(*delta), (const uint4 *)d_data.p, (const uint4 *)d_w.p, n, (uint4 *)d_out.p,
                    ctx->d_te0, s);
    GC_HIP(hipGetLastError());
    GC_HIP(hipMemcpyAsync(out, d_out.p, n * 32, hipMemcpyDeviceToHost, s));
    GC_HIP(hipStreamSynchronize(s));
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_cot_receive_unpad(gc_ctx *ctx, const gc_label *seed, const uint8_t *flags, const gc_label *sent,
                         gc_label *result, size_t n) try {
    if (!ctx || !seed || (n && (!flags || !sent || !result))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GC_HIP(hipSetDevice(ctx->device));
    DevBuf d_f, d_s, d_r;
    GC_HIP(d_f.alloc(n));
    GC_HIP(d_s.alloc(n * 32));
    GC_HIP(d_r.alloc(n * 16));
    hipStream_t s = ctx->stream;
    GC_HIP(hipMemcpyAsync(d_f.p, flags, n, hipMemcpyHostToDevice, s));
    GC_HIP(hipMemcpyAsync(d_s.p, sent, n * 32, hipMemcpyHostToDevice, s));
    GC_HIP(hipMemcpyAsync(d_r.p, result, n * 16, hipMemcpyHostToDevice, s));
    launch_cot_recv(to_u4(*seed), (const uint8_t *)d_f.p, (const uint4 *)d_s.p, (uint4 *)d_r.p, n, ctx->d_te0, s);
    GC_HIP(hipGetLastError());
    GC_HIP(hipMemcpyAsync(result, d_r.p, n * 16, hipMemcpyDeviceToHost, s));
    GC_HIP(hipStreamSynchronize(s));
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

// device-resident forms of the two COT pad loops: device pointers, asynchronous on the ctx stream, no staging
int gc_cot_send_pads_dev(gc_ctx *ctx, const gc_label *seed, const gc_label *delta, const void *d_data,
                         const void *d_wires, size_t n, void *d_out) {
    if (!ctx || !seed || !delta || (n && (!d_data || !d_wires || !d_out))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    GC_HIP(hipSetDevice(ctx->device));
    launch_cot_send(to_u4(*seed), to_u4(*delta), (const uint4 *)d_data, (const uint4 *)d_wires, n, (uint4 *)d_out,
                    ctx->d_te0, ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

int gc_cot_receive_unpad_dev(gc_ctx *ctx, const gc_label *seed, const void *d_flags, const void *d_sent,
                             void *d_result, size_t n) {
    if (!ctx || !seed || (n && (!d_flags || !d_sent || !d_result))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    GC_HIP(hipSetDevice(ctx->device));
    launch_cot_recv(to_u4(*seed), (const uint8_t *)d_flags, (const uint4 *)d_sent, (uint4 *)d_result, n, ctx->d_te0,
                    ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

// ---- ROT (ot/rot.go:132-202): the pad loops of ROT.Send / ROT.Receive --------------------------------------------
// Sender: wires[j] = {H_j(data_j), H_j(data_j ^ Delta)} with H_j(x) = x ^ AES_{key_j}(x), key_j = BE(Label{j, 0} ^ seed)
// (mitccrh.go:70-89, a fresh MITCCRH per Send: rot.go:143).  Receiver: result[j] = H_j(result[j]).  The same kernels as
// the COT pads (k_cot_dual: one lane = one OT, the per-OT key schedule running one round ahead of the state).
int gc_rot_send_dev(gc_ctx *ctx, const gc_label *seed, const gc_label *delta, const void *d_data, size_t n,
                    void *d_wires_out) {
    if (!ctx || !seed || !delta || (n && (!d_data || !d_wires_out))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    GC_HIP(hipSetDevice(ctx->device));
    launch_cot_send(to_u4(*seed), to_u4(*delta), (const uint4 *)d_data, nullptr, n, (uint4 *)d_wires_out, ctx->d_te0,
                    ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

int gc_rot_receive_dev(gc_ctx *ctx, const gc_label *seed, void *d_result, size_t n) {
    if (!ctx || !seed || (n && !d_result)) return GC_E_ARG;
    if (n == 0) return GC_OK;
    GC_HIP(hipSetDevice(ctx->device));
    launch_mitccrh(to_u4(*seed), 0, (uint4 *)d_result, n, 1, ctx->d_te0, ctx->stream);
    GC_HIP(hipGetLastError());
    return GC_OK;
}

int gc_rot_send(gc_ctx *ctx, const gc_label *seed, const gc_label *delta, const gc_label *data, size_t n,
                gc_wire *wires_out) try {
    if (!ctx || !seed || !delta || (n && (!data || !wires_out))) return GC_E_ARG;
    if (n == 0) return GC_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GC_HIP(hipSetDevice(ctx->device));
    DevBuf d_data, d_out;
    GC_HIP(d_data.alloc(n * 16));
    GC_HIP(d_out.alloc(n * 32));
    GC_HIP(hipMemcpyAsync(d_data.p, data, n * 16, hipMemcpyHostToDevice, ctx->stream));
    launch_cot_send(to_u4(*seed), to_u4(*delta), (const uint4 *)d_data.p, nullptr, n, (uint4 *)d_out.p, ctx->d_te0, ctx->stream);
    GC_HIP(hipGetLastError());
    GC_HIP(hipMemcpyAsync(wires_out, d_out.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GC_OK;
} catch (...) {
    return gc::on_exception();
}

int gc_rot_receive(gc_ctx *ctx, const gc_label *seed, gc_label *result, size_t n) {
    return gc_mitccrh_hash(ctx, seed, 0, result, n, 1);
}

}  // extern "C"
