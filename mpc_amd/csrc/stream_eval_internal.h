// stream_eval_internal.h — the streaming evaluator's state, shared by stream_eval.cpp (the C ABI, the parser, the host-side
// skeleton match, scheduling) and stream_eval_dev.cpp (the device-side match of whole read buffers).
#pragma once

#include "stream_internal.h"
#include "stream_skel.h"

namespace gcs {
struct EvalDev;  // stream_eval_dev.cpp
}

using namespace gcs;  // (internal header of the engine's own translation units)

struct gc_stream_eval {
    gc_ctx *ctx = nullptr;
    std::vector<uint8_t> key;
    int rounds = 0;
    uint32_t *d_rk = nullptr;  // expanded key on the device (step groups)
    DevStore store;  // StreamEval.wires (global store), device-resident
    CircCache cache;
    size_t cache_gates = 0, cache_budget = kCacheGatesDefault;
    uint64_t tick = 0;
    // step groups (see the head of this file): small blocks that share no global wire are evaluated by ONE launch sequence
    std::vector<std::unique_ptr<Slot>> slots;
    GroupWindow win;
    CtxQueue ctxq;
    DeepLanes deep;
    std::vector<gc_label> rows_scratch;  // table rows of a small block while it is parsed
    uint64_t n_groups = 0, n_group_blocks = 0;
    FuseStats fuse;  // chain fusion (stream_fuse.cpp)
    bool use_deps = deps_wanted();  // units that wait inside a launch (stream_internal.h: kUnitDeps)
    std::vector<uint32_t> wiring_scratch;
    std::vector<uint32_t> io_host;  // indices of this block's inputs, then of its global outputs (0xffffffff: superseded)
    uint32_t *d_io = nullptr;
    size_t io_cap = 0;
    // per-circuit scratch, kept across calls: last writer of every tmp / global wire with a generation stamp
    std::vector<uint64_t> last_t, last_w;  // per tmp / global wire: generation stamp << 32 | current id (one load per look-up)
    uint32_t gen = 0;
    std::vector<gc_gate> gates;       // only materialised for a circuit the cache does not know
    std::vector<CircKey> keys;        // the block's gates as packed records {in0, in1, out, op}
    std::vector<uint64_t> dst_pack;   // per gate: destination index | tmp flag << 32 (parser scratch, kept across calls)
    std::vector<uint32_t> in_idx, id_of;
    // by (ngates, ntmp); a few byte layouts per key, most recently matched first (behind pointers: the order changes often)
    std::unordered_map<uint64_t, std::vector<std::unique_ptr<EvalSkel>>> skels;
    std::vector<uint32_t> gf_ids, wr_ids;   // scratch: the block's global ids by field / the wires it writes
    EvalSkel rec;                           // skeleton of the block being parsed (kept when the parse succeeds)
    bool use_skels = true;                  // GC_STREAM_NO_SKELETON (read at creation): every block is parsed
    SkelPool pool;                          // helper threads of the skeleton match (started by the first big block)
    uint64_t n_parsed = 0, n_matched = 0;
    size_t skel_bytes = 0;                  // reference bytes held by skels (capped: the blocks are the peer's data)
    // table rows of the block being parsed, in pinned memory (true asynchronous H2D); two buffers: the copy of block k
    // may still be in flight while block k + 1 is parsed
    // (kEvalRing buffers: a block's set is re-used once the pass of the block kEvalRing calls ago has run)
    static constexpr uint32_t kEvalRing = 4;
    gc_label *slab_pin[kEvalRing] = {};
    size_t slab_cap[kEvalRing] = {};
    hipEvent_t slab_ev[kEvalRing] = {};
    // ... and its wire maps (inputs, then global outputs): pinned staging + a device copy per ring entry — from pageable
    // memory the upload is synchronous with the stream (the host would wait for the previous block's kernels at every
    // block: the evaluator ran at host time PLUS GPU time per big block), and one device copy would be overwritten under
    // the pass still reading it
    uint32_t *io_pin[kEvalRing] = {};
    uint32_t *io_dev[kEvalRing] = {};
    size_t io_pin_cap[kEvalRing] = {}, io_dev_cap[kEvalRing] = {};
    uint32_t slab_turn = 0;
    // The uploads of a big block (rows, wire maps) run on a stream of their own, under the kernels of the block before:
    // up_ev[i] = this entry's uploads done (the ctx stream waits for it); ring_batch[i] = the pooled batch whose table
    // buffer they went into (an upload into it waits for slab_ev[i]: the pass that last read it); held = the batch of the
    // last block, kept out of the pool until the next block has taken its own (two passes in flight, two table buffers)
    hipStream_t up_stream = nullptr;
    hipEvent_t up_ev[kEvalRing] = {};
    gc_batch *ring_batch[kEvalRing] = {};
    gc_circ *held_circ = nullptr;
    gc_batch *held = nullptr;
    StageProf prof;
    uint64_t n_blocks_total = 0;
    gcs::EvalDev *dev = nullptr;  // device-side match of the blocks of a read buffer (stream_eval_dev.cpp; created on first use)
};


namespace gcs {

// What eval_schedule needs to know about a block whose circuit has been found (parsed, matched on the host, or matched on the
// device): the circuit, its wires (e->io_host: inputs, then global outputs with 0xffffffff for superseded ones; e->wr_ids: the
// wires it writes), and where its table rows are.
struct BlockIn {
    CircEntry *ent;
    uint32_t ngates, nin, nout;
    size_t nrows, pos;          // rows; bytes of the block (what *consumed becomes)
    bool small_block;
    uint32_t sb;                // big block: its entry of the pinned ring (slab = e->slab_pin[sb], filled unless rows_from)
    gc_label *slab;             // rows in host order (small block: e->rows_scratch) — unless they are still in buf:
    const EvalSkel *rows_from;  // a matched small block: rows_from->copy_rows(buf, ...) moves them once, where they go
    const uint8_t *buf;
    // ... or in device memory (stream_eval_dev.cpp: the block was matched there): the block's body, the skeleton's row offsets
    // (device arrays), the chunk of the peer's stream they are in — a small block's rows are then gathered into its job's
    // table array by the group's launch sequence, no byte of them passes through the host
    const uint8_t *d_block = nullptr;
    const uint32_t *d_row_off = nullptr;
    uint32_t chunk = 0;
};
int eval_rows_buffer(gc_stream_eval *e, uint32_t ngates, bool *small_block, uint32_t *sb, gc_label **slab);
int eval_schedule(gc_stream_eval *e, const BlockIn &in, size_t *consumed);
int eval_block(gc_stream_eval *e, uint32_t ngates, uint32_t ntmp, uint32_t nwires, const uint8_t *buf, size_t len, size_t *consumed);
uint32_t stream_max_wires();
// the bookkeeping of a block that equals skeleton sk byte-wise, from its global ids (ids[f] = the id in field f, as the device
// read them): do they repeat in the skeleton's pattern, are they below nwires?  Fills e->io_host / e->wr_ids.
bool eval_adopt_ids(gc_stream_eval *e, const EvalSkel &sk, const uint32_t *ids, uint32_t nwires);

// ---- device-side match (stream_eval_dev.cpp) ----------------------------------------------------------------------------------
// every complete OpCircuit block at the head of buf[0, len) that the DEVICE recognises (or the host parser takes when the device
// does not know it): *pos = bytes taken, *n = blocks; what is left — a cut-off block, another operation — is the caller's
int evdev_blocks(gc_stream_eval *e, const uint8_t *buf, size_t len, size_t *pos, uint32_t *n);
// a skeleton has been made / is going away: its device copy
void evdev_add(gc_stream_eval *e, EvalSkel *sk);
void evdev_drop(gc_stream_eval *e, EvalSkel *sk);
// after a group has been launched: the chunks of the peer's stream its jobs gather their rows from are in use by that launch
void evdev_ref(gc_stream_eval *e, uint32_t chunk);  // a job that gathers its rows from this chunk has been queued
void evdev_launched(gc_stream_eval *e, Slot &g, uint32_t slot_index);
void evdev_free(gc_stream_eval *e);
void evdev_stats(const gc_stream_eval *e, uint64_t *blocks, uint64_t *fallbacks);

}  // namespace gcs
