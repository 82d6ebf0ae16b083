// kernels.h — launch wrappers of the HIP kernels (implemented in gc_kernels.hip / ot_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.h"

namespace gc {

// Placement of a [row][instance] array of 16-byte labels in HBM.
//   level layout  (schedule 0): row-major, one row = bstride instances  -> a wave reads 64 consecutive
//                               instances of one wire = 1 KiB coalesced
//   tiled layout  (schedule 1): instances are grouped in tiles of TI = 1<<ti_log2; one tile's rows are
//                               contiguous ([tile][row][TI]) so that the workgroup owning the tile
//                               keeps its whole working set in one L2-friendly region
struct Layout {
    uint32_t ti_log2;    // 31 for the level layout (a single tile)
    uint32_t ti_mask;    // (1 << ti_log2) - 1
    size_t row_stride;   // elements between consecutive rows
    size_t tile_stride;  // elements between consecutive tiles (0 for the level layout)
#if defined(__HIPCC__)
    __host__ __device__
#endif
    size_t at(size_t row, uint32_t inst) const {
        return (size_t)(inst >> ti_log2) * tile_stride + row * row_stride + (inst & ti_mask);
    }
};

// Geometry of one instance batch on the device.
struct BatchGeom {
    uint32_t batch;    // instances
    uint32_t bstride;  // instances allocated (batch rounded up to 64 / to the tile size)
    uint32_t lg;       // level kernels: log2 of the per-block instance tile (<= 8)
    uint32_t yblocks;  // level kernels: instance blocks of 256 (1 when batch <= 256)
    uint32_t ti_log2;  // fused kernels: log2 of the instances per workgroup tile
    uint32_t ntiles;
    bool lds_wires;    // fused kernels: wire labels live in LDS (hash-phase schedule)
    Layout lw;         // wire labels  [nslots][...]
    Layout lt;         // table rows   [slab_rows][...]
};
// schedule 0 = level layout, 1 = tiled layout
// nls: live-label high-water mark of the LDS schedule (0xffffffff: does not fit -> global-memory wires)
// nls: live labels of the LDS plan that will run (0xffffffff: none); flat: size them for the flattened kernels
BatchGeom make_geom(uint32_t batch, int schedule, uint32_t nslots, uint32_t slab_rows, uint32_t nls,
                    uint32_t max_ti_log2 = 6, bool flat = false, uint32_t ustride = 0);

struct LevelArgs {
    const GateDesc *descs;  // device, already offset to the step
    uint32_t count, nonfree, out_slot0;
    uint4 *W;               // wire labels
    const uint4 *R;         // garbler: [bstride]
    uint4 *T;               // tables
    const uint32_t *rk;     // device round keys (big-endian words)
    const uint32_t *te0;    // device Te0
    int rounds;
};

void launch_garble_level(const LevelArgs &a, const BatchGeom &g, hipStream_t s);
void launch_eval_level(const LevelArgs &a, const BatchGeom &g, hipStream_t s);

// Fused schedule: ONE launch walks all levels; a workgroup owns a tile of instances and
// synchronises its own waves between levels (no inter-workgroup dependency at all).
struct FusedArgs {
    const GateDesc *descs;
    const Step *steps;  // device copy of Plan::levels
    uint32_t nsteps, ninputs;
    uint4 *W;
    const uint4 *R;
    uint4 *T;
    const uint32_t *rk;
    const uint32_t *te0;
    int rounds;
    uint64_t *prof;  // optional [ntiles][8] cycle breakdown (debug), nullptr in production
    bool narrow_col; // every level fits one column-sliced pass (fused_col_form_fits): k_garble_col / k_eval_col
};
bool fused_col_form_fits(const Step *levels, uint32_t nsteps, uint32_t ti_log2);
void launch_garble_fused(const FusedArgs &a, const BatchGeom &g, hipStream_t s);
void launch_eval_fused(const FusedArgs &a, const BatchGeom &g, hipStream_t s);
// ONE instance with wires in HBM: one launch per level, pass k of the level = workgroup k (lanes along the gates);
// levels = host copy of the step array
uint32_t level1_passes(const Step &st, bool eval);
void launch_levels1(bool eval, const FusedArgs &a, const Step *levels, hipStream_t s);
// the same walk as ONE launch: kCoopGroups workgroups of one XCD (one L2) walk the levels together, a counter in that L2
// is the barrier between levels (fused_kernels.hip: k_garble_coop).  ctl lives in device memory; a pass leaves the counters
// at zero for the next one (no memset between passes: a fill kernel plus its gaps was 10 us of a 250 us step).
struct CoopCtl {
    uint32_t count;        // arrivals, monotonic over the barriers of one launch; zero between launches
    uint32_t error;        // a workgroup gave up waiting (never seen after a passed self-test; stays up until reported)
    uint32_t bad;          // self-test: values that did not arrive, workgroups on another XCD
    uint32_t ticks;        // self-test: s_memtime ticks of its 128 barriers
    uint32_t left;         // workgroups that finished the pass (the last one zeroes count and left)
    uint32_t repaired;     // passes that lost a workgroup and were done again by k_coop_repair (gc_kernels.hip)
    uint32_t *host_err;    // a word of pinned host memory (device-visible) that a workgroup giving up raises as well: the
                           // host reads it without a copy per pass
    uint32_t passes;       // passes launched so far (counted by the last workgroup to leave)
    uint32_t drop_at;      // testing (GC_COOP_FORCE_TIMEOUT = n): in the n-th pass one workgroup takes no part: the others'
                           // bounded wait runs out, exactly as if it had not become resident in time
    uint32_t scratch[64];  // self-test
};
constexpr uint32_t kCoopGroups = 32;
// Label exchange with a device-resident wire store (the streaming passes), done by the cooperative kernel itself instead of
// a kernel before and one behind it: W[i] = store[in_idx[i]] for the ninputs input slots, a barrier, the levels, a barrier,
// store[out_idx[j]] = W[out_slots[j]] (out_idx 0xffffffff: not stored).
struct StoreXchg {
    uint4 *store;
    const uint32_t *in_idx;
    const uint32_t *out_slots;
    const uint32_t *out_idx;
    uint32_t nout;
};
// x: device address of the pass's StoreXchg, or nullptr (kernel arguments stay in SGPRs for the whole pass and the round
// keys already fill them: the exchange by value cost 17 more spilled SGPRs and 30 us of a 200 us pass)
// seq: cooperative passes launched on this ctl before this one (= ctl->passes when the pass starts).  The launch carries a
// STAND-BY workgroup on the same XCD: it sleeps until the pass has ended (ctl->passes > seq) and returns — unless the pass
// raised ctl->error (a workgroup's bounded wait ran out): then it does the whole pass again by itself, level by level behind a
// barrier of its own (slow, correct: the per-gate code of the level launches, level_gate.h), including the exchange with the
// wire store that the failed pass skipped, and clears the flag.  The kernel ends when the stand-by does, so a lost workgroup
// costs time, never a result: what follows on the stream sees the labels and table rows of a good pass.
void launch_coop(bool eval, const FusedArgs &a, CoopCtl *ctl, const StoreXchg *x, uint32_t seq, hipStream_t s);
void launch_coop_selftest(CoopCtl *ctl, hipStream_t s);

// Fused schedule with LDS-resident wires (fused_lds_kernels.hip)
struct FusedLdsArgs {
    const FDesc *descs;
    const uint32_t *gslot;
    const Step *steps;
    const Chunk *chunks;
    const uint16_t *in_lds;
    uint32_t nsteps, nchunks, ninputs, nls;
    uint4 *W;
    const uint4 *R;
    uint4 *T;
    const uint32_t *rk;
    const uint32_t *te0;
    int rounds;
    bool store_all;  // also write every wire label to the global array (debug / Garbled.Wires)
    uint64_t *prof;
};
size_t fused_lds_bytes(uint32_t nls, uint32_t ti_log2);
hipError_t launch_fused_lds(bool eval, const FusedLdsArgs &a, const BatchGeom &g, hipStream_t s);

// Fused, LDS-resident, flattened schedule (fused_flat_kernels.hip): the production path
struct FusedFlatArgs {
    const uint32_t *prog;  // Plan::fl_prog
    const FUnit *units;
    const uint32_t *hgslot, *ogslot;
    const uint16_t *in_lds;
    uint32_t nunits, ninputs, nls;  // nls = Plan::n_flat_slots (zero slot included)
    uint32_t ustride;               // Plan::fl_unit_stride: uint4 per LDS stage buffer (largest unit of the circuit)
    uint4 *W;
    const uint4 *R;
    uint4 *T;
    const uint32_t *rk;
    const uint32_t *te0;
    int rounds;
    uint64_t *prof;  // optional [ntiles][16] cycle breakdown (debug), nullptr in production
    const uint4 *rnd;  // garble: draw R and the input labels from this stream inside the kernel (no init launch)
    bool has_or;       // the circuit has OR gates (selects the kernel build with the OR paths)
};
size_t fused_flat_bytes(uint32_t nls, uint32_t ti_log2, uint32_t ustride);
hipError_t launch_fused_flat(bool eval, const FusedFlatArgs &a, const BatchGeom &g, hipStream_t s);

// What one workgroup of the flattened kernels needs.  The batch launches pass ONE of these as the kernel argument (every
// workgroup = one tile of the same circuit); the streaming engine's step groups pass an ARRAY in device memory, one
// record per workgroup (launch_fused_flat_jobs): workgroup j runs ONE instance of job j's circuit — its own plan, wire
// and table arrays — and takes its input labels straight from the stream's device-resident wire store
// (W[i] = store[in_idx[i]]), so that independent SSA-step circuits of a stream run side by side in a single launch.
struct FlatJob {
    const uint4 *prog;
    const FUnit *units;
    const uint32_t *hgslot, *ogslot;
    const uint16_t *in_lds;
    uint32_t nunits, ninputs, ti_log2, zslot;
    uint32_t ustride;  // uint4 per stage buffer
    size_t w_tile, t_tile;
    uint4 *W;
    const uint4 *R;
    uint4 *T;
    const uint32_t *rk;
    const uint32_t *te0;
    uint64_t *prof;    // debug profile of the batch kernels; JOB launches (never profiled): the unit's DfBlock, or nullptr
    const uint4 *rnd;  // garbler only: the caller's random stream (nullptr: R / input labels are already in place)
    uint4 *Rout;       // garbler only: R of every instance, for the later passes of the pipeline
    uint32_t batch;
    uint32_t pad_;
    uint4 *store;                // job launches: the stream's wire store ...
    const uint32_t *in_idx;      // ... the store index of every input wire (Get through in[]) ...
    const uint32_t *out_slots;   // ... and, for output k, its wire slot in W and its store index (Set through out[];
    const uint32_t *out_idx;     //     0xffffffff: not stored), written back by the workgroup itself when it is done
    uint32_t nout, pad2_;
};
// d_jobs: device array of records (ti_log2 = 0, batch = 1, rnd = prof = nullptr); d_first == nullptr: workgroup u runs record
// u; else the records d_first[u] .. d_first[u] + d_jobs[d_first[u]].pad_ one after the other (pad2_ = a record's position in
// that run: 0 loads the AES table) — several for a chain of dependent steps without a merged plan (stream_fuse.cpp);
// lds_bytes = the largest fused_flat_bytes(nls, 0, ustride) of the group; has_or: some job's circuit has an OR gate; rounds:
// 10 / 12 / 14 (one key per stream).
// d_sync != nullptr (with d_first): units of the launch DEPEND on one another.  The block (device memory, uploaded with these
// words) is [0] a ticket counter, zero; [1] nunits (bit 31: full agent-scope fences around the waits, a cross-check); [2..3] device address of a pinned host word raised to 3 when a wait runs
// out (0: none); [kSyncHead ..) one done-flag per unit, zero; then nunits + 1 offsets into the list behind them: the units
// that unit u waits for are list[off[u]] .. list[off[u + 1]) and ALL have a lower index than u.  A workgroup takes the next
// ticket as its unit number (so every unit it may wait for has started: no wait can starve, whatever order the hardware
// dispatches workgroups in), waits for the done-flags of its list, runs its records, raises its own flag.
constexpr uint32_t kSyncHead = 4;
// Dataflow between the launch units of a stream ACROSS launches (round 6, GC_STREAM_DATAFLOW, garbler): the wire store carries,
// beside every label, how often the wire has been WRITTEN (ver) and READ (rd) by launch units so far; the host keeps the same
// counts in launch order (= program order for every pair of units that share a wire) and tells each unit, per input, the version
// it must find before it loads the label, and per output the version and the number of reads the wire must have reached before
// it may be overwritten, and the version the write makes.  Launches then need no order between one another: a unit starts when
// ITS operands exist.  FlatJob::prof of a job record points at this block (device memory): the header, then ver_in[nin], then
// {prev, reads, next}[nout].  Every wait is bounded by polls; a wait that runs out raises *host_err to 3 and goes on.
struct DfBlock {
    uint32_t *ver, *rd;    // per global wire (as the store)
    uint32_t *host_err;    // pinned word of the ctx (gc_ctx_err_word), device address
    uint32_t nin, nout;
    uint32_t pad_[2];
};
static_assert(sizeof(DfBlock) == 40 || sizeof(DfBlock) == 48, "DfBlock layout");
// ... and out-of-order ISSUE on top of it (GC_STREAM_DATAFLOW=3): the units of every launch are PUBLISHED into a ring of the
// stream, in program order, and the workgroups of ANY launch claim the lowest unclaimed one, run it, and claim again until
// nothing is left — a unit whose own launch sits behind a waiting kernel in its in-order queue is run by a workgroup of another
// launch.  Claims are in ticket order, so a unit only ever waits for units that are already running: no wait can starve, whatever
// is resident.  A unit is `more + 1` job records in a row (a chain without a merged plan: stream_fuse.cpp); when it is done it
// counts itself in done[group] (the serialiser of that group waits for the count: k_pool_wait) and in done_total (joins).
constexpr uint32_t kPoolRing = 8192;    // > the steps a stream may have in flight (kMaxPending)
constexpr uint32_t kPoolGroups = 8192;
struct PoolEntry {
    const FlatJob *rec;
    uint32_t more, group;
};
struct PoolCtl {
    uint32_t head, tail;      // tickets claimed / published
    uint32_t done_total, pad_;
    uint32_t *host_err;
    // GC_STREAM_DATAFLOW=4: the workgroups are PERSISTENT — launched once, they wait (sleeping polls) for units to be published
    // instead of leaving when there is none, until the host raises `stop` behind its last publication (a join: big steps,
    // read-backs, the stream's end); units then never wait for a workgroup of their own launch to reach the head of its queue
    uint32_t persist, stop;
    uint32_t done[kPoolGroups];
    PoolEntry ring[kPoolRing];
};
hipError_t launch_fused_flat_pool(int rounds, PoolCtl *ctl, uint32_t nworkers, size_t lds_bytes, hipStream_t s);
void launch_pool_publish(PoolCtl *ctl, const PoolEntry *d_entries, uint32_t first_ticket, uint32_t n, hipStream_t s);
void launch_pool_wait(const uint32_t *d_counter, uint32_t want, uint32_t *d_host_err, hipStream_t s);
void launch_pool_stop(PoolCtl *ctl, hipStream_t s);
hipError_t launch_fused_flat_jobs(bool eval, int rounds, bool has_or, const FlatJob *d_jobs, const uint32_t *d_first, uint32_t nunits,
                                  size_t lds_bytes, hipStream_t s, uint32_t *d_sync = nullptr);

// rnd [batch][1+ninputs] big-endian label bytes -> R[inst] (S bit set) and W[w][inst]
void launch_init_garble(const uint4 *rnd, uint32_t ninputs, uint4 *W, uint4 *R, const BatchGeom &g, hipStream_t s);

// generic [instance][n] <-> [row][instance] movers (16-byte elements, LDS-tiled transpose).
// slots == nullptr means row j = slot0 + j.  lay = placement of the device array.
// mode: 0 = labels, 1 = wires {L0, L0^R} (gather only)
void launch_gather(const uint4 *W, const Layout &lay, uint32_t inst0, const uint32_t *slots, uint32_t slot0,
                   uint32_t n, const uint4 *R, int mode, uint4 *dst, size_t dst_stride_elems, uint32_t count,
                   hipStream_t s);
void launch_scatter(const uint4 *src, size_t src_stride_elems, uint32_t n, const uint32_t *slots, uint32_t slot0,
                    uint4 *W, const Layout &lay, uint32_t inst0, uint32_t count, hipStream_t s);

// one-instance batch <-> device-resident wire store (streaming): W[i] = store[idx[i]];  store[idx[j]] = W[slots[j]]
// (idx 0xffffffff: skip)
void launch_store_gather(uint4 *W, const uint4 *store, const uint32_t *idx, uint32_t n, hipStream_t s);
void launch_store_scatter(uint4 *store, const uint4 *W, const uint32_t *slots, const uint32_t *idx, uint32_t n, hipStream_t s);

void launch_select_inputs(uint4 *We, const uint4 *Wg, const uint4 *R, const uint8_t *bits, uint32_t ninputs,
                          const BatchGeom &g, hipStream_t s);
void launch_decode(const uint4 *Wg, const uint4 *R, const uint4 *We, const uint32_t *out_slots, uint32_t noutputs,
                   uint8_t *bits_out, uint32_t *mismatch, const BatchGeom &g, hipStream_t s);
// dst [n][bstride] <- W[slots[j]][*]
void launch_gather_rows(const uint4 *W, const uint32_t *slots, uint32_t n, uint4 *dst, const BatchGeom &g,
                        hipStream_t s);

// Table egress / ingest in the wire format of circuit.Garbler / circuit.Evaluator (garbler.go:69-82,
// evaluator.go:40-66): per instance  BE32(#gates) | per gate BE32(#rows) + rows as BE(D0)||BE(D1).
// ops[g] / rows[g] are in ORIGINAL gate order.  stride = bytes between instances in the byte buffer.
void launch_tables_egress(const uint4 *T, const Layout &lt, const uint8_t *ops, const uint32_t *row_of_gate,
                          uint32_t ngates, uint32_t batch, uint8_t *out, size_t stride, hipStream_t s);
// *bad is incremented for every header that does not match (gate count, per-gate row count)
void launch_tables_ingest(uint4 *T, const Layout &lt, const uint8_t *ops, const uint32_t *row_of_gate,
                          uint32_t ngates, uint32_t batch, const uint8_t *in, size_t stride, uint32_t *bad,
                          hipStream_t s);

// sha2pc's headerless table encoding (sha2pc/encoding.go:363-411): rows in gate order, BE(D0)||BE(D1) each
void launch_slab_be(uint4 *T, const Layout &lt, uint32_t rows, uint32_t batch, uint8_t *buf, size_t stride, bool ingest,
                    hipStream_t s);

// ---- OT kernels (ot_kernels.hip) -------------------------------------------------------------
// IKNP OT extension, fused (iknp_fused_kernels.hip): column AES-128-CTR PRG + u-matrix / delta fold + createLabels.
// rk0/rk1: [128][44] expanded column keys (big-endian words); pos0: bytes every column stream has already
// produced; n OTs in chunks of 512.
//   recv: u_out = PRG(g0)^PRG(g1)^choice bytes (bbuf: choice bits packed LSB first, 64 bytes per chunk),
//         labels = transpose(PRG(g0))
//   send: labels = transpose(PRG(g0) ^ (delta.Bit(col) ? u_in : 0))
hipError_t launch_iknp_fused(bool recv, const uint32_t *rk0, const uint32_t *rk1, uint64_t pos0, size_t n,
                             const uint8_t *bbuf, const uint8_t *u_in, uint4 delta, uint8_t *u_out, uint4 *labels,
                             const uint32_t *te0, hipStream_t s);
void launch_pack_bits(const uint8_t *b, size_t n, uint8_t *out, hipStream_t s);
// bit-COT helpers: choice words -> per-chunk choice bytes (whole 64-bit words only, iknp.go:583-597); bit 0 of every label
void launch_fold_choice_words(const uint64_t *choices, size_t n, uint64_t *bbuf, hipStream_t s);
void launch_pack_label_bit0(const uint4 *labels, size_t n, uint64_t *result, hipStream_t s);
// KOS check accumulators: acc[0..3] ^= XOR chi_i * v_i (256 bit), acc[4..5] ^= XOR_{bits_i} chi_i
void launch_kos_accumulate(const uint32_t *rk, uint64_t idx0, const uint4 *v, const uint8_t *bits, size_t n,
                           unsigned long long *acc, const uint32_t *te0, hipStream_t s);
// blk[j*h + t] ^= AES_{key(gid0+j)}(blk[j*h+t])
void launch_mitccrh(uint4 seed, uint64_t gid0, uint4 *blks, size_t n, uint32_t h, const uint32_t *te0,
                    hipStream_t s);
void launch_cot_send(uint4 seed, uint4 delta, const uint4 *data, const uint4 *wires /*[n][2]*/, size_t n, uint4 *out,
                     const uint32_t *te0, hipStream_t s);
void launch_cot_recv(uint4 seed, const uint8_t *flags, const uint4 *sent, uint4 *result, size_t n,
                     const uint32_t *te0, hipStream_t s);

}  // namespace gc
