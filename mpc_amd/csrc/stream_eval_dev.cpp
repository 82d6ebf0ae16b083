// stream_eval_dev.cpp — the streaming evaluator's blocks recognised ON THE DEVICE.
//
// The evaluator's host thread used to touch every byte of the peer's stream twice and a half: a masked compare of each block
// with its byte skeleton (stream_skel.h), then one shuffle per table row into the group's upload region — 14 us per 88 KB block
// of the Ed25519-shaped program, 2.3 GB in all: memory speed of one core, and the bound of the evaluator.
// None of that is host work.  gc_stream_eval_blocks hands a whole read buffer over; here it goes to the GPU as it is (DMA in
// pieces: straight from the caller's buffer when that is pinned — gc_host_alloc / gc_host_register —, else through pinned
// staging) and the host only GUESSES where its blocks are: header -> the skeletons of this (gates, tmp wires) key, most
// recently matched first -> the first one behind whose length another header follows.  That is 24 bytes per block.  Whether
// the guess holds is the device's to say, for all blocks of a piece side by side (k_eval_verify: workgroup b = block b, the
// masked compare with the skeleton, the block's global wire ids read at the skeleton's field offsets), into pinned host
// memory, piece after piece behind the DMA.  Nothing is scheduled on a guess: the host follows the verdicts as they arrive —
// ids -> repeat pattern check -> reads / writes of the block -> group / chain / lane, exactly as after a host-side match
// (eval_schedule) — except that the rows stay where they are: the group's launch sequence gathers them from the device copy
// of the stream into the job's table array (stream_group.cpp: k_rows_gather).
//
// A block the device refuses (the guess was wrong, or its ids repeat in another pattern) goes to the host's matcher / parser
// (stream_eval.cpp: eval_block), and so does whatever the guessing cannot place: a block met for the first time, a block cut
// off by the end of the buffer, another operation.  That is also where new skeletons come from; their device copies are
// made here (evdev_add).  The peer's bytes are untrusted: a verdict is only asked for bytes inside [0, len), the kernel trusts
// only the skeleton table the host built, sizes and ids are checked by the host as before (tests/hostile_fuzz.py drives this
// path too).
#include "stream_eval_internal.h"

#include <immintrin.h>

namespace gcs {

namespace {

constexpr uint32_t kMaxRecs = 1u << 16;        // blocks of one read buffer
constexpr uint32_t kMaxIds = 1u << 23;         // global ids of one read buffer
constexpr size_t kDevSkelBytes = (size_t)1 << 30;
constexpr size_t kMinDevBuffer = 64 << 10;     // shorter read buffers stay on the host (a launch and a DMA are ~30 us)
constexpr size_t kPieceFirst = (size_t)512 << 10, kPieceMax = (size_t)4 << 20;  // the buffer goes up in pieces, each followed by
                                               // the verdicts on the blocks that end in it: a short one first (the host has
                                               // something to do after 10 us), then doubling (a copy + a launch cost the host ~20 us)

struct DevSkel {       // device mirror of an EvalSkel
    uint32_t nbytes, ngf;
    uint32_t nuniq;    // distinct global wires the block names
    uint32_t mate;     // the next skeleton of the same SHAPE (the same bytes, mask, fields, rows: only the ids' repeat pattern
                       // differs — an adder bound to another constant); a ring through all of them
    const uint8_t *bytes, *mask;
    const uint32_t *gf_off;    // [ngf] byte offset of every global-id field | 1 << 31 for 4-byte ids
    const uint32_t *gf_canon;  // [ngf] the first field that names the same wire
    const uint32_t *uniq_f;    // [nuniq] the fields that name a wire for the first time, in stream order
};
struct VerifyItem {    // host -> device: "the block whose body starts here is skeleton `skel`"
    uint64_t body;     // offset of the block's body (behind its 20-byte header) in the buffer
    uint32_t skel, ids_off;
};
struct VerifyCtl {
    uint32_t ndone;    // items with a verdict (device -> host, system scope)
    uint32_t pad_[3];
};
struct VerifyArgs {
    const uint8_t *buf;
    const VerifyItem *items;  // this batch
    const DevSkel *skels;
    uint32_t *ids;
    uint32_t *ok;             // this batch: 1 = the block equals the skeleton outside its global ids and rows
    uint32_t *counter;        // device word, zero: workgroups of this batch that are done
    uint32_t nitems, ndone_after;
    VerifyCtl *ctl;
};

typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint32_t v4u_unaligned __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t be32_at(const uint8_t *p) { return __builtin_bswap32(*(const u32_unaligned *)p); }

constexpr uint32_t kUniqLds = 8192;  // distinct wires of a block the verdict can check for distinctness (more: the host's path)

__device__ __forceinline__ uint32_t field_id(const uint8_t *blk, uint32_t off) {
    const uint8_t *q = blk + (off & 0x7fffffffu);
    return (off >> 31) ? be32_at(q) : ((uint32_t)q[0] << 8) | q[1];
}

// Workgroup b = block b of the batch: (1) does it equal the guessed skeleton outside its global ids and rows (the masked
// compare), (2) which skeleton of that shape has the block's repeat pattern of ids — every field names the wire its first
// occurrence names, the first occurrences name different wires (what canon_of checks on the host, stream_eval.cpp) — and
// (3) the distinct ids themselves, in order of first occurrence: all the host needs to know the block's reads and writes.
// ok[b] = 0: not this shape / no pattern fits (the host's matcher takes the block), else 1 + the skeleton.
__global__ __launch_bounds__(512) void k_eval_verify(VerifyArgs a) {
    __shared__ uint32_t uid[kUniqLds];
    const uint32_t tid = threadIdx.x;
    const VerifyItem it = a.items[blockIdx.x];
    DevSkel k = a.skels[it.skel];
    const uint8_t *blk = a.buf + it.body;
    // (block ^ reference) & mask over the whole block: 16 bytes per lane and array at a time, four rounds in flight
    uint32_t diff = 0;
    const uint32_t n16 = k.nbytes >> 4;
    for (uint32_t i = tid; i < n16; i += 4 * 512) {
        v4u acc = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t j = i + u * 512;
            if (j < n16) acc |= (((const v4u_unaligned *)blk)[j] ^ ((const v4u *)k.bytes)[j]) & ((const v4u *)k.mask)[j];
        }
        diff |= acc.x | acc.y | acc.z | acc.w;
    }
    for (uint32_t b = (n16 << 4) + tid; b < k.nbytes; b += 512) diff |= (uint32_t)((blk[b] ^ k.bytes[b]) & k.mask[b]);
    uint32_t verdict = 0;
    if (!__syncthreads_or((int)(diff != 0))) {
        uint32_t s = it.skel;
        for (uint32_t tries = 0; tries < 64; tries++) {  // the guessed skeleton first, then its mates
            int bad = k.nuniq > kUniqLds;
            if (!bad) {
                for (uint32_t f = tid; f < k.ngf; f += 512) {
                    const uint32_t c = k.gf_canon[f];
                    if (c != f && field_id(blk, k.gf_off[f]) != field_id(blk, k.gf_off[c])) bad = 1;
                }
                for (uint32_t u = tid; u < k.nuniq; u += 512) uid[u] = field_id(blk, k.gf_off[k.uniq_f[u]]);
            }
            bad = __syncthreads_or(bad);
            if (!bad) {  // the first occurrences name different wires?  (a few hundred of them: all pairs)
                const uint32_t n = k.nuniq;
                for (uint32_t u = tid; u < n; u += 512) {
                    const uint32_t v = uid[u];
                    for (uint32_t w = 0; w < u; w++) bad |= uid[w] == v;
                }
                bad = __syncthreads_or(bad);
            }
            if (!bad) {
                for (uint32_t u = tid; u < k.nuniq; u += 512) a.ids[it.ids_off + u] = uid[u];
                verdict = 1 + s;
                break;
            }
            s = k.mate;
            if (s == it.skel) break;
            k = a.skels[s];
            __syncthreads();
        }
    }
    if (tid == 0) {
        a.ok[blockIdx.x] = verdict;
        __threadfence_system();
        // the last workgroup of the batch tells the host (batches run one after the other: ndone only grows)
        if (__hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == a.nitems - 1) {
            __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(&a.ctl->ndone, a.ndone_after, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace

struct EvalDev {
    int state = 0;  // 0: not set up, 1: ready, -1: off
    // the skeleton table: device array + its host copy, the host skeleton behind every entry, the key hash
    std::vector<DevSkel> skels;
    std::vector<EvalSkel *> host_of;
    std::vector<void *> owned;  // device arrays of the entries (freed with the evaluator: a walk may still be reading them)
    DevSkel *d_skels = nullptr;
    size_t d_skels_cap = 0, held = 0;
    bool dirty = true;
    // the host's guesses and the device's verdicts, pinned and device-visible
    VerifyCtl *ctl = nullptr, *d_ctl = nullptr;
    VerifyItem *items = nullptr, *d_items = nullptr;
    uint32_t *ok = nullptr, *d_ok = nullptr;
    uint32_t *ids = nullptr, *d_ids = nullptr;
    uint32_t *d_counter = nullptr;
    hipStream_t vstream = nullptr;        // the verdicts run here, the copies on the evaluator's upload stream: piece k + 1 goes
    std::vector<hipEvent_t> piece_ev;     // up while piece k's blocks are looked at
    std::vector<EvalSkel *> guess;  // the skeleton behind every item of the buffer in hand (nullptr: it has gone meanwhile)
    std::vector<uint32_t> guess_bytes;  // ... and its length
    // chunks of the peer's stream on the device: alive while a queued job gathers its rows from them
    struct Chunk {
        uint8_t *d = nullptr;
        size_t cap = 0;
        uint32_t refs = 0;   // jobs queued and not launched yet
        bool walking = false;
        std::vector<std::pair<uint32_t, uint64_t>> users;  // (slot, launch number) of the launches that read it
    };
    std::vector<Chunk> chunks;
    uint8_t *stage = nullptr;  // pinned staging for a caller's pageable buffer: filled piece by piece by a few copier threads
    size_t stage_cap = 0;      // (one core copies ~12 GB/s; the stream of the Ed25519-shaped program wants 20) while the pieces
    CopyPool copier;           // already there go up and are looked at
    std::unique_ptr<std::atomic<uint32_t>[]> piece_copies;
    size_t piece_copies_cap = 0;
    uint64_t n_blocks = 0, n_fallback = 0;
};

namespace {

bool dev_setup(gc_stream_eval *e) {
    if (!e->dev) e->dev = new (std::nothrow) EvalDev;
    EvalDev *d = e->dev;
    if (!d) return false;
    if (d->state != 0) return d->state > 0;
    d->state = -1;
    if (std::getenv("GC_STREAM_NO_DEVICE_MATCH") || !e->use_skels) return false;
    if (hipSetDevice(e->ctx->device) != hipSuccess) return false;
    hipError_t er = hipHostMalloc((void **)&d->ctl, sizeof(VerifyCtl), hipHostMallocDefault);
    if (er == hipSuccess) er = hipHostMalloc((void **)&d->items, (size_t)kMaxRecs * sizeof(VerifyItem), hipHostMallocDefault);
    if (er == hipSuccess) er = hipHostMalloc((void **)&d->ok, (size_t)kMaxRecs * sizeof(uint32_t), hipHostMallocDefault);
    if (er == hipSuccess) er = hipHostMalloc((void **)&d->ids, (size_t)kMaxIds * sizeof(uint32_t), hipHostMallocDefault);
    if (er == hipSuccess) er = hipHostGetDevicePointer((void **)&d->d_ctl, d->ctl, 0);
    if (er == hipSuccess) er = hipHostGetDevicePointer((void **)&d->d_items, d->items, 0);
    if (er == hipSuccess) er = hipHostGetDevicePointer((void **)&d->d_ok, d->ok, 0);
    if (er == hipSuccess) er = hipHostGetDevicePointer((void **)&d->d_ids, d->ids, 0);
    if (er == hipSuccess) er = hipMalloc((void **)&d->d_counter, 64);
    if (er == hipSuccess) er = hipMemset(d->d_counter, 0, 64);
    if (er == hipSuccess && !e->up_stream) er = hipStreamCreateWithFlags(&e->up_stream, hipStreamNonBlocking);
    if (er == hipSuccess) er = hipStreamCreateWithFlags(&d->vstream, hipStreamNonBlocking);
    if (er != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    d->state = 1;
    // skeletons made before the first read buffer came this way
    for (auto &kv : e->skels)
        for (auto &sk : kv.second)
            if (sk->dev_index < 0) evdev_add(e, sk.get());
    return true;
}

// the chunks nobody reads any more go back to the ctx's buffer lists
void chunks_sweep(gc_stream_eval *e) {
    EvalDev *d = e->dev;
    for (auto &c : d->chunks) {
        if (!c.d || c.refs || c.walking) continue;
        bool busy = false;
        for (const auto &u : c.users) {
            const Slot &g = *e->slots[u.first];
            // (a slot that has been given back or launched again since: that launch is long done)
            if (g.kind == Slot::kGroup && g.launched && g.launch_no == u.second && g.error == GC_OK && hipEventQuery(g.done) != hipSuccess) busy = true;
        }
        (void)hipGetLastError();
        if (busy) continue;
        gc::ctx_buf_put(e->ctx, false, c.d, c.cap);
        c = EvalDev::Chunk{};
    }
}

}  // namespace

void evdev_add(gc_stream_eval *e, EvalSkel *sk) {
    EvalDev *d = e->dev;
    if (!d || d->state <= 0 || sk->dev_index >= 0 || sk->nbytes >= 0x7fffffffu) return;
    const size_t ngf = sk->gf_off.size();
    const size_t need = 2 * sk->nbytes + ngf * 12 + sk->row_off.size() * 4 + 64;
    if (d->held + need > kDevSkelBytes) return;
    if (hipSetDevice(e->ctx->device) != hipSuccess) return;
    // the distinct wires of the block in order of first occurrence, and which of them every input / output is
    std::vector<uint32_t> uniq_f, u_of(ngf, 0);
    for (uint32_t f = 0; f < ngf; f++) {
        if (sk->gf_canon[f] == f) {
            u_of[f] = (uint32_t)uniq_f.size();
            uniq_f.push_back(f);
        } else {
            u_of[f] = u_of[sk->gf_canon[f]];
        }
    }
    sk->in_u.resize(sk->in_gf.size());
    sk->out_u.resize(sk->out_gf.size());
    for (size_t k = 0; k < sk->in_gf.size(); k++) sk->in_u[k] = u_of[sk->in_gf[k]];
    for (size_t k = 0; k < sk->out_gf.size(); k++) sk->out_u[k] = u_of[sk->out_gf[k]];
    uint8_t *db = nullptr, *dm = nullptr;
    uint32_t *dg = nullptr, *dr = nullptr, *dc = nullptr, *du = nullptr;
    const size_t nb = (sk->nbytes + 15) & ~(size_t)15;
    hipError_t er = hipMalloc((void **)&db, nb + 16);
    if (er == hipSuccess) er = hipMalloc((void **)&dm, nb + 16);
    if (er == hipSuccess) er = hipMalloc((void **)&dg, ngf * 4 + 4);
    if (er == hipSuccess) er = hipMalloc((void **)&dc, ngf * 4 + 4);
    if (er == hipSuccess) er = hipMalloc((void **)&du, uniq_f.size() * 4 + 4);
    if (er == hipSuccess) er = hipMalloc((void **)&dr, sk->row_off.size() * 4 + 4);
    if (er == hipSuccess) er = hipMemcpy(db, sk->bytes.data(), sk->nbytes, hipMemcpyHostToDevice);
    if (er == hipSuccess) er = hipMemcpy(dm, sk->mask.data(), sk->nbytes, hipMemcpyHostToDevice);
    if (er == hipSuccess && ngf) er = hipMemcpy(dg, sk->gf_off.data(), ngf * 4, hipMemcpyHostToDevice);
    if (er == hipSuccess && ngf) er = hipMemcpy(dc, sk->gf_canon.data(), ngf * 4, hipMemcpyHostToDevice);
    if (er == hipSuccess && !uniq_f.empty()) er = hipMemcpy(du, uniq_f.data(), uniq_f.size() * 4, hipMemcpyHostToDevice);
    if (er == hipSuccess && !sk->row_off.empty()) er = hipMemcpy(dr, sk->row_off.data(), sk->row_off.size() * 4, hipMemcpyHostToDevice);
    if (er != hipSuccess) {
        (void)hipGetLastError();
        for (void *p : {(void *)db, (void *)dm, (void *)dg, (void *)dc, (void *)du, (void *)dr})
            if (p) (void)hipFree(p);
        return;
    }
    for (void *p : {(void *)db, (void *)dm, (void *)dg, (void *)dc, (void *)du, (void *)dr}) d->owned.push_back(p);
    d->held += need;
    sk->dev_index = (int)d->skels.size();
    sk->d_row_off = dr;
    sk->n_uniq = (uint32_t)uniq_f.size();
    d->host_of.push_back(sk);
    d->skels.push_back(DevSkel{(uint32_t)sk->nbytes, (uint32_t)ngf, (uint32_t)uniq_f.size(), (uint32_t)sk->dev_index, db, dm, dg, dc, du});
    d->dirty = true;
}

void evdev_drop(gc_stream_eval *e, EvalSkel *sk) {
    EvalDev *d = e->dev;
    if (!d || sk->dev_index < 0) return;
    d->host_of[(size_t)sk->dev_index] = nullptr;  // (its device arrays stay until the evaluator goes: a verdict may be under way)
    sk->dev_index = -1;
    for (auto &g : d->guess)  // (a buffer in hand may have guessed it for blocks still to come: those take the host's path)
        if (g == sk) g = nullptr;
    d->dirty = true;  // (it leaves the ring of its shape with the next upload of the table)
}

void evdev_ref(gc_stream_eval *e, uint32_t chunk) {
    if (e->dev && chunk < e->dev->chunks.size()) e->dev->chunks[chunk].refs++;
}

void evdev_launched(gc_stream_eval *e, Slot &g, uint32_t slot_index) {
    EvalDev *d = e->dev;
    if (!d) return;
    for (uint32_t c : g.chunk_refs) {
        EvalDev::Chunk &ch = d->chunks[c];
        if (ch.refs) ch.refs--;
        if (ch.users.empty() || ch.users.back() != std::make_pair(slot_index, g.launch_no)) ch.users.emplace_back(slot_index, g.launch_no);
    }
    g.chunk_refs.clear();
}

void evdev_stats(const gc_stream_eval *e, uint64_t *blocks, uint64_t *fallbacks) {
    if (blocks) *blocks = e->dev ? e->dev->n_blocks : 0;
    if (fallbacks) *fallbacks = e->dev ? e->dev->n_fallback : 0;
}

void evdev_free(gc_stream_eval *e) {
    EvalDev *d = e->dev;
    if (!d) return;
    d->copier.wait();
    d->copier.stop();
    if (e->ctx) (void)hipSetDevice(e->ctx->device);
    if (e->up_stream) (void)hipStreamSynchronize(e->up_stream);
    if (d->vstream) {
        (void)hipStreamSynchronize(d->vstream);
        (void)hipStreamDestroy(d->vstream);
    }
    for (hipEvent_t ev : d->piece_ev) (void)hipEventDestroy(ev);
    for (void *p : d->owned) (void)hipFree(p);
    for (auto &c : d->chunks)
        if (c.d) gc::ctx_buf_put(e->ctx, false, c.d, c.cap);
    if (d->stage) gc::ctx_buf_put(e->ctx, true, d->stage, d->stage_cap);
    if (d->d_skels) (void)hipFree(d->d_skels);
    if (d->d_counter) (void)hipFree(d->d_counter);
    if (d->ctl) (void)hipHostFree(d->ctl);
    if (d->items) (void)hipHostFree(d->items);
    if (d->ok) (void)hipHostFree(d->ok);
    if (d->ids) (void)hipHostFree(d->ids);
    delete d;
    e->dev = nullptr;
}

namespace {

// the skeleton table on the device as the host has it
int table_upload(gc_stream_eval *e) {
    EvalDev *d = e->dev;
    if (!d->dirty) return GC_OK;
    hipStream_t st = e->up_stream;
    {  // the rings of the skeletons of one shape (the live ones: a dropped skeleton is a ring of its own, nobody guesses it)
        std::unordered_map<uint64_t, uint32_t> last;  // shape class -> the latest entry seen
        std::unordered_map<uint64_t, uint32_t> first;
        for (uint32_t i = 0; i < d->skels.size(); i++) {
            d->skels[i].mate = i;
            const EvalSkel *sk = d->host_of[i];
            if (!sk) continue;
            const uint64_t cls = (sk->shape ^ sk->key * 0x9e3779b97f4a7c15ull) + sk->nbytes;
            auto it = last.find(cls);
            if (it == last.end()) {
                first[cls] = i;
            } else {
                d->skels[it->second].mate = i;
                d->skels[i].mate = first[cls];
            }
            last[cls] = i;
        }
    }
    GC_HIP(hipStreamSynchronize(d->vstream));  // (no verdict may be under way on the table that changes)
    if (d->skels.size() > d->d_skels_cap) {
        if (d->d_skels) (void)hipFree(d->d_skels);
        d->d_skels = nullptr;
        d->d_skels_cap = d->skels.size() + d->skels.size() / 2 + 64;
        GC_HIP(hipMalloc((void **)&d->d_skels, d->d_skels_cap * sizeof(DevSkel)));
    }
    GC_HIP(hipMemcpyAsync(d->d_skels, d->skels.data(), d->skels.size() * sizeof(DevSkel), hipMemcpyHostToDevice, st));
    GC_HIP(hipStreamSynchronize(st));  // (a pageable source: the vector may change before an asynchronous copy has read it)
    d->dirty = false;
    return GC_OK;
}

inline uint32_t be32h(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

}  // namespace

int evdev_blocks(gc_stream_eval *e, const uint8_t *buf, size_t len, size_t *pos_out, uint32_t *n_out) {
    *pos_out = 0;
    *n_out = 0;
    if (len < kMinDevBuffer || len >= ((size_t)1 << 32) || e->skels.empty() || !dev_setup(e)) return GC_OK;
    EvalDev *d = e->dev;
    gc_ctx *ctx = e->ctx;
    GC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = e->up_stream;
    // ---- where are the blocks?  A guess per block: the most recently matched skeleton of its key that fits what is here and
    //      behind whose length another operation's header (or the end of the buffer) follows.  Stops at the first block it
    //      cannot place.
    e->prof.start();
    d->guess.clear();
    d->guess_bytes.clear();
    size_t pos = 0;
    uint32_t nitems = 0, ids_used = 0;
    while (len - pos >= 20 && be32h(buf + pos) == 1 /* OpCircuit */ && nitems < kMaxRecs) {
        const uint32_t ngates = be32h(buf + pos + 8), ntmp = be32h(buf + pos + 12);
        auto it = e->skels.find(((uint64_t)ngates << 32) | ntmp);
        if (it == e->skels.end()) break;
        const size_t body = pos + 20, avail = len - body;
        EvalSkel *pick = nullptr;
        for (auto &sk : it->second) {
            if (sk->dev_index < 0 || sk->nbytes > avail) continue;
            const size_t nxt = body + sk->nbytes;
            if (len - nxt >= 4 && be32h(buf + nxt) > 64) continue;  // (no operation code follows: not this layout)
            pick = sk.get();
            break;
        }
        if (!pick || ids_used + pick->gf_off.size() > kMaxIds) break;
        d->items[nitems] = VerifyItem{(uint64_t)body, (uint32_t)pick->dev_index, ids_used};
        d->guess.push_back(pick);
        d->guess_bytes.push_back((uint32_t)pick->nbytes);
        ids_used += (uint32_t)pick->gf_off.size();
        nitems++;
        pos = body + pick->nbytes;
    }
    e->prof.lap(StageProf::kGuess);
    if (nitems < 4) return GC_OK;  // (nothing to gain: the host's loop takes the buffer)
    const size_t upto = pos;       // bytes the guesses cover
    int rc = table_upload(e);
    if (rc != GC_OK) return rc;
    chunks_sweep(e);
    // ---- the buffer goes to the device, piece by piece, every piece followed by the verdicts on the blocks that end in it
    uint32_t ci = 0;
    while (ci < d->chunks.size() && d->chunks[ci].d) ci++;
    if (ci == d->chunks.size()) d->chunks.emplace_back();
    {
        void *p = nullptr;
        size_t cap = 0;
        GC_HIP(gc::ctx_buf_get(ctx, false, upto + 16, &p, &cap));
        EvalDev::Chunk &c = d->chunks[ci];
        c.d = (uint8_t *)p, c.cap = cap, c.refs = 0, c.walking = true;
        c.users.clear();
    }
    uint8_t *d_buf = d->chunks[ci].d;
    const bool pinned = gc_host_is_pinned(buf) != 0;
    if (!pinned && grow_pin(ctx, &d->stage, &d->stage_cap, upto) != hipSuccess) {
        d->chunks[ci].walking = false;
        return GC_E_NOMEM;
    }
    d->ctl->ndone = 0;
    hipError_t er = hipSuccess;
    if (!pinned) {  // a pageable buffer: through pinned staging (the DMA cannot read it in place) — all pieces to the copier threads
        size_t npieces = 0, piece = kPieceFirst;
        for (size_t off = 0; off < upto; off += std::min(piece, upto - off), piece = std::min(piece * 2, kPieceMax)) npieces++;
        if (npieces > d->piece_copies_cap) {
            d->piece_copies.reset(new std::atomic<uint32_t>[npieces + 8]);
            d->piece_copies_cap = npieces + 8;
        }
        piece = kPieceFirst;
        size_t k = 0;
        for (size_t off = 0, nb = 0; off < upto; off += nb, piece = std::min(piece * 2, kPieceMax), k++) {
            nb = std::min(piece, upto - off);
            d->piece_copies[k].store(0, std::memory_order_relaxed);
            d->copier.submit(d->stage + off, buf + off, nb, &d->piece_copies[k]);
        }
    }
    {
        uint32_t first = 0, pi = 0;  // (pi: events in use — one per piece that is followed by verdicts)
        size_t piece = kPieceFirst, k = 0;
        for (size_t off = 0, nb = 0; off < upto && er == hipSuccess; off += nb, piece = std::min(piece * 2, kPieceMax), k++) {
            nb = std::min(piece, upto - off);
            const uint8_t *src = buf + off;
            if (!pinned) {
                while (d->piece_copies[k].load(std::memory_order_acquire) != 0) _mm_pause();  // (this piece is in the staging)
                src = d->stage + off;
            }
            er = hipMemcpyAsync(d_buf + off, src, nb, hipMemcpyHostToDevice, st);
            uint32_t last = first;  // items [first, last) end inside what has gone up so far
            while (last < nitems && d->items[last].body + d->guess_bytes[last] <= off + nb) last++;
            if (er == hipSuccess && last > first) {
                if (pi >= d->piece_ev.size()) {
                    hipEvent_t ev = nullptr;
                    if ((er = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) break;
                    d->piece_ev.push_back(ev);
                }
                er = hipEventRecord(d->piece_ev[pi], st);
                if (er == hipSuccess) er = hipStreamWaitEvent(d->vstream, d->piece_ev[pi], 0);
                if (er != hipSuccess) break;
                VerifyArgs a{d_buf, d->d_items + first, d->d_skels, d->d_ids, d->d_ok + first, d->d_counter, last - first, last, d->d_ctl};
                hipLaunchKernelGGL(k_eval_verify, dim3(last - first), dim3(512), 0, d->vstream, a);
                er = hipGetLastError();
                first = last;
                pi++;
            }
        }
    }
    if (er != hipSuccess) {
        if (!pinned) d->copier.wait();  // (the copier threads read the caller's buffer: not beyond this call)
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(d->vstream);
        d->chunks[ci].walking = false;
        set_error("gc_stream_eval_blocks (device copy / verdicts)", er);
        return GC_E_HIP;
    }
    e->prof.lap(StageProf::kOther);
    // ---- follow the verdicts as they arrive
    uint32_t next = 0, n = 0;
    uint64_t spins = 0;
    pos = 0;
    bool resync = false;  // a block turned out to have another length than its guess: what follows is not where it was guessed
    while (next < nitems && rc == GC_OK && !resync) {
        const uint32_t have = __atomic_load_n(&d->ctl->ndone, __ATOMIC_ACQUIRE);
        if (next >= have) {
            _mm_pause();
            if ((++spins & 0xfffff) == 0 && hipStreamQuery(d->vstream) != hipErrorNotReady && __atomic_load_n(&d->ctl->ndone, __ATOMIC_ACQUIRE) <= next) {
                (void)hipGetLastError();
                set_error("gc_stream_eval_blocks (verdicts)", hipErrorUnknown);
                rc = GC_E_HIP;
            }
            continue;
        }
        e->prof.lap(StageProf::kWait);
        for (; next < have && rc == GC_OK && !resync; next++) {
            const VerifyItem r = d->items[next];
            if (next + 2 < nitems) {  // (the header and the ids of the block after next: written by the peer / the device, cold)
                const VerifyItem &nx = d->items[next + 2];
                __builtin_prefetch(buf + nx.body - 20);
                for (uint32_t l = 0; l < 16; l++) __builtin_prefetch(d->ids + nx.ids_off + 16 * l);
            }
            EvalSkel *sk = d->guess[next];
            const uint8_t *hdr = buf + r.body - 20;
            const uint32_t ngates = be32h(hdr + 8), ntmp = be32h(hdr + 12), nwires = be32h(hdr + 16);
            // the block is the peer's: the same bounds as eval_block's, before anything is sized by its header
            if ((uint64_t)ntmp > 64ull * ngates + (1u << 20) || nwires > stream_max_wires()) {
                rc = GC_E_ARG;
                break;
            }
            size_t used = 0;
            // the verdict names the skeleton of the guessed one's shape whose repeat pattern the block's ids have, and brings the
            // distinct ids: the block's reads and writes follow
            const uint32_t *d_row_off = sk ? sk->d_row_off : nullptr;
            bool adopted = false;
            const uint32_t verdict = d->ok[next];
            if (sk && verdict && ngates && verdict - 1 < d->host_of.size()) {
                EvalSkel *o = d->host_of[verdict - 1];
                if (o && o->ent && o->key == sk->key && o->nbytes == sk->nbytes) {
                    const uint32_t *uid = d->ids + r.ids_off;
                    bool in_range = true;
                    for (uint32_t u = 0; u < o->n_uniq; u++) in_range = in_range && uid[u] < nwires;
                    if (in_range) {
                        const uint32_t nin = o->nin, nout = o->nout;
                        e->io_host.resize((size_t)nin + nout + 1);
                        for (uint32_t k = 0; k < nin; k++) e->io_host[k] = uid[o->in_u[k]];
                        e->wr_ids.resize(nout);
                        for (uint32_t k = 0; k < nout; k++) {
                            e->wr_ids[k] = uid[o->out_u[k]];
                            e->io_host[nin + k] = o->out_live[k] ? e->wr_ids[k] : 0xffffffffu;
                        }
                        adopted = true;
                        if (e->win.rec.size() >= e->store.host.size()) {
                            e->win.prefetch(e->io_host.data(), nin);
                            e->win.prefetch(e->wr_ids.data(), nout);
                        }
                        if (o != sk) {  // most recently matched first: the next block of this key is guessed to be this layout
                            auto it = e->skels.find(sk->key);
                            if (it != e->skels.end())
                                for (size_t si = 1; si < it->second.size(); si++)
                                    if (it->second[si].get() == o) {
                                        std::rotate(it->second.begin(), it->second.begin() + (long)si, it->second.begin() + (long)si + 1);
                                        break;
                                    }
                            sk = o;
                        }
                    }
                }
            }
            e->prof.lap(StageProf::kAdopt);
            if (adopted) {
                e->store.ensure(nwires);  // InitCircuit(numWires, numTmpWires)
                BlockIn in{};
                in.ent = sk->ent, in.ngates = ngates, in.nin = sk->nin, in.nout = sk->nout, in.nrows = sk->nrows, in.pos = sk->nbytes;
                in.buf = buf + r.body;
                if ((rc = eval_rows_buffer(e, ngates, &in.small_block, &in.sb, &in.slab)) != GC_OK) break;
                if (in.small_block) {
                    in.rows_from = sk;  // (used only if the block ends up on a path that wants its rows on the host)
                    in.d_block = d_buf + r.body, in.d_row_off = d_row_off, in.chunk = ci;
                } else {
                    sk->copy_rows(buf + r.body, in.slab);  // a block of many gates: the pinned ring, as after a host match
                }
                sk->ent->last_use = ++e->tick;
                rc = eval_schedule(e, in, &used);
                e->n_matched++;
                d->n_blocks++;
            } else {  // not what was guessed, or its ids repeat in another pattern: the host's matcher / parser takes this block
                rc = eval_block(e, ngates, ntmp, nwires, buf + r.body, len - r.body, &used);
                d->n_fallback++;
                if (rc == GC_E_ROWS) {  // (cut off by the end of the buffer after all: the caller's loop says so)
                    rc = GC_OK;
                    resync = true;
                    break;
                }
                if (rc == GC_OK && used != d->guess_bytes[next]) resync = true;
            }
            if (rc != GC_OK) break;
            pos = r.body + used;
            n++;
        }
    }
    if (rc != GC_OK || resync) (void)hipStreamSynchronize(d->vstream);  // (verdicts still under way write into the arrays of the next buffer)
    d->chunks[ci].walking = false;  // (from here on the chunk lives as long as queued jobs and running launches refer to it)
    *pos_out = pos;
    *n_out = n;
    return rc;
}

}  // namespace gcs
