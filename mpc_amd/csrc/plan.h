// plan.h — host-side levelised re-encoding of a circuit (no HIP types here).
//
// The reference executes gates serially (circuit/garble.go:285-299, circuit/eval.go:28-112) and
// derives two running counters from that order: the hash tweak `id` (AND +2, OR/INV +1:
// garble.go:357-359,419-420,451-452) and the slab offset (AND 2, OR 3, INV 1: garble.go:199-211,
// 296-298).  The plan freezes both per gate, renames wires to single-assignment slots and buckets
// gates by dependency level (== Circuit.AssignLevels(TargetYao), circuit.go:206-254), so that a
// level can run as one data-parallel launch and still produce the serial loop's exact bytes.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/gcengine.h"

namespace gc {

// status of the exception in flight (called from the catch (...) of an extern "C" entry point: no C++ exception may
// cross the C ABI — under cgo it would end in std::terminate): std::bad_alloc / length_error -> GC_E_NOMEM, anything
// else GC_E_HIP; the text goes to gc_last_error().  Defined in engine.cpp.
int on_exception() noexcept;

// One gate as the kernels see it (16 bytes, one s_load_dwordx4 when wave-uniform).
struct GateDesc {
    uint32_t in0;     // wire slot of input 0
    uint32_t in1;     // wire slot of input 1 (== in0 for INV)
    uint32_t tweak;   // reference `id` at this gate
    uint32_t row_op;  // first slab row | op << 29
};
static_assert(sizeof(GateDesc) == 16, "GateDesc must be 16 bytes");

constexpr uint32_t kOpShift = 29;
constexpr uint32_t kRowMask = (1u << kOpShift) - 1;

// A launch unit: descs [first, first+count) are mutually independent; the first `nonfree`
// of them (AND, OR, INV — sorted to the front) need the AES tables.
struct Step {
    uint32_t first;
    uint32_t count;
    uint32_t nonfree;
    uint32_t n_and, n_or, n_inv;  // sorted in this order at the front; nonfree = n_and + n_or + n_inv
};

// Descriptor of the LDS-resident fused schedule (16 bytes).  Wire labels live in LDS slots that are
// recycled as soon as the last reader has run (linear-scan allocation over the step order).
struct FDesc {
    uint32_t lin;      // LDS slot of input 0 | LDS slot of input 1 << 16
    uint32_t lout;     // LDS slot of the output | flags << 16 (bit 16: also store to the global wire array)
    uint32_t tweak;
    uint32_t row_op;   // first slab row | op << 29
};
static_assert(sizeof(FDesc) == 16, "FDesc must be 16 bytes");
constexpr uint32_t kFStoreGlobal = 1u << 16;
constexpr uint32_t kFLevelStart = 1u << 17;  // first descriptor of an XOR sub-level (flat XOR streams)

// Descriptor staging unit of the LDS schedule: a run of whole steps whose descriptors (<= kChunkDescs)
// and step records (<= kChunkSteps) are copied into LDS in one go while the previous chunk executes.
// A chunk normally is one hash phase plus the XOR sub-levels that follow it.
constexpr uint32_t kChunkDescs = 768;
constexpr uint32_t kChunkSteps = 32;
struct Chunk {
    uint32_t first_step, nsteps;
    uint32_t first_desc, ndesc;  // ndesc > kChunkDescs only for a single over-wide step (read from HBM directly)
};

// ---- flattened schedule (fused_flat_kernels.hip; the production path) ---------------------------------
// Free-XOR makes every XOR/XNOR output a GF(2) combination of labels that exist when the preceding hash
// phase has finished ("chunk-entry" labels: input wires, table-producing gates, XOR outputs of earlier
// chunks).  Instead of walking the XOR gates level by level — a serial, latency-bound chain of ~8 dependent
// LDS round trips per hash phase — the plan expands each XOR output that somebody actually needs (a hashed
// gate, a later chunk, a circuit output) into its term list and the kernels compute all of them in ONE
// parallel step per hash phase; intermediate XOR wires are never materialised.  A term list longer than
// kFlatMaxTerms becomes two levels: blocks of kFlatMaxTerms chunk-entry terms are summed by XOuts of round 1 that stand
// for no wire (shared by content between the values of an accumulator chain), the value is the XOR of the block sums in
// round 2 (plan.cpp: build_flat); only a list of more than kFlatMaxTerms block sums falls back to materialising the
// operands and opening another round.
// (garble.go:331-351 / eval.go:49-51 compute the same labels gate by gate.)
constexpr uint32_t kFlatMaxTerms = 32;
// One XOR work item (24 bytes, self-contained: a lane needs ONE LDS round trip for it and one for its labels).
// An output with more than 8 terms is spread over 2 or 4 consecutive XOuts ("parts", index of the first a multiple
// of the count): every lane reads at most 8 labels and the leader (part 0) collects the partial sums of the lanes
// TI, 2 TI, 3 TI further on with DPP row shifts (parts x TI <= 16).
struct XOut {
    uint16_t t[8];       // LDS slots of the terms, unused ones = the zero slot
    uint16_t out;        // LDS slot of the result (leader only)
    uint16_t flags;      // kXo*
    uint16_t n;          // number of terms of this part (terms 4..7 are only read when n > 4)
    uint16_t pad_;
};
static_assert(sizeof(XOut) == 24, "XOut must be 24 bytes");
constexpr uint16_t kXoStore = 1;  // also store to the global wire array (circuit output)
constexpr uint16_t kXoRpar = 2;   // garbler: odd number of XNORs in the expansion -> XOR R once
constexpr uint16_t kXoJoin2 = 4;  // leader of 2 parts
constexpr uint16_t kXoJoin4 = 8;  // leader of 4 parts
constexpr uint16_t kXoPart = 16;  // parts 1..3: partial sum only, nothing is written
// A unit is what the kernels stage into LDS in one go: [hash descriptors][XOuts], executed as
// hash part -> barrier -> XOR part -> barrier.  Capacities chosen so that a unit is <= 928 uint4 (one per
// thread) and two buffers cost 29 KiB of LDS.
constexpr uint32_t kUHash = 256, kUOuts = 448;
constexpr uint32_t kUnit16 = kUHash + kUOuts * 24 / 16;  // 928
struct FUnit {           // 48 bytes
    uint32_t off16, n16;             // position / size of the unit in Plan::fl_prog (uint4 units)
    uint32_t n_and, n_or, n_inv;     // hash descriptors (16 B each) at the start of the unit, in this order
    uint32_t nout;                   // XOuts
    uint32_t outs_off16;             // relative to the unit start
    uint32_t xparts;                 // largest part count of an XOut in this unit (1, 2 or 4)
    uint32_t hfirst, ofirst;         // index of the first hash desc / XOut in fl_hgslot / fl_ogslot
    uint32_t pad_[2];
};
static_assert(sizeof(FUnit) == 48, "FUnit must be 48 bytes");

struct Plan {
    gc_plan_info info{};
    // original gate order
    std::vector<uint32_t> level_of_gate, tweak_of_gate, row_of_gate /* ngates+1 */, slot_of_gate;
    // execution order: desc k writes slot ninputs + k
    std::vector<GateDesc> descs;
    std::vector<uint32_t> gate_of_desc;
    std::vector<Step> levels;  // schedule 0: one step per dependency level
    // wire id -> slot holding its final value (inputs: identity); 0xffffffff if never written
    std::vector<uint32_t> slot_of_wire;
    std::vector<uint32_t> out_slots;  // slots of wires [nwires-noutputs, nwires)

    // ---- hash-phase schedule with LDS-resident wires (fused kernels) ----
    // Steps alternate between "hash phases" (all table-producing gates of one non-free depth) and the
    // XOR sub-levels that follow them; 89 hash phases instead of 290 hash-carrying levels for aes_128.
    std::vector<FDesc> fdescs;          // execution order
    std::vector<uint32_t> fgslot;       // global wire slot written by fdescs[k]
    std::vector<Step> fsteps;
    std::vector<Chunk> fchunks;
    // flattened schedule (empty / n_flat_slots == 0xffffffff when the live set does not fit 16-bit slots)
    std::vector<uint32_t> fl_prog;      // unit images, 16-byte aligned pieces (see FUnit)
    std::vector<FUnit> fl_units;
    std::vector<uint32_t> fl_hgslot, fl_ogslot;  // global wire slot of every hash desc / XOut
    std::vector<uint16_t> fl_in_lds;    // LDS slot of every input wire (0xffff: never read)
    uint32_t n_flat_slots = 0xffffffffu;  // live labels incl. the zero slot (= slot n_flat_slots - 1)
    uint32_t n_flat_outs = 0, n_flat_terms = 0, n_flat_steps = 0;
    bool flat_late = false;             // the flattened schedule runs every hashed gate as late as its consumers allow (plan.cpp: build_flat)
    uint32_t fl_unit_stride = 0;        // uint4 per LDS stage buffer = the largest unit (<= kUnit16)
    uint32_t fl_max_parts = 1;          // largest XOut part count: a tile may hold at most 16 / fl_max_parts instances
    std::vector<uint16_t> in_lds;       // LDS slot of every input wire (0xffff: never read)
    uint32_t n_lds_slots = 0;           // high-water mark of live labels
    uint32_t n_hash_phases = 0;
    // The flattened plan is 70 % of the build time (~0.5 us per gate) and some users never need it (ONE instance of a wide
    // circuit runs level launches, kernels.h): build_plan(..., defer_flat) keeps what build_flat needs instead
    // (gate ops and the resolved producers of pass 1) and finish_flat() builds it on first demand.
    bool flat_built = false;
    std::vector<uint8_t> lazy_ops;                          // op of every gate
    std::vector<uint32_t> lazy_src0, lazy_src1, lazy_cur;   // producer of input 0 / 1 of every gate, of every wire
    uint32_t passes_garble = 0, passes_eval = 0;            // 1024-lane passes of all levels, one instance (kernels.h)
};

// returns GC_OK or GC_E_GATE / GC_E_WIRE / GC_E_ARG
// seg_first / nseg: optional, ascending gate indices at which the hash tweak `id` starts over at 0 — the gate list is a
// CHAIN of streamed circuits fused into one planned job (stream_fuse.cpp) and Streaming.Garble restarts the tweak per circuit
// (circuit/stream_garble.go:174); table rows keep counting through the whole list.
int build_plan(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs, uint32_t noutputs,
               Plan *out, bool defer_flat = false, const uint32_t *seg_first = nullptr, uint32_t nseg = 0);
// builds the flattened schedule of a plan made with defer_flat (no-op once built); not thread-safe: callers serialise
void finish_flat(Plan *p);
// ONE instance of this circuit is better off as one launch per level with the level's passes spread over workgroups
// (its levels average >= 2.5 passes of 1024 lanes) than as a single workgroup walking it
bool wide_for_one_instance(const Plan &p, bool eval);

// plaintext walk of the flattened unit program (host-side self-check, see plan.cpp)
int simulate_flat(const Plan &p, const uint8_t *in_bits, uint8_t *out_bits);

// LDS geometry of the flattened kernels that the planner needs to know (fused_flat_kernels.hip asserts the same values)
constexpr uint32_t kFlatStageOff16 = 65536 / 16 + 16;  // uint4 in front of the stage buffers: the AES table + column keys
constexpr size_t kFlatLdsBytes = 160 * 1024;

}  // namespace gc

struct gc_plan {
    gc::Plan p;
};
