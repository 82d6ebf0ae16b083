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

// ---- staggered half-tile schedule (fused_lds2_kernels.hip) ----------------------------------------------
// Same hash-phase order, re-cut into "stagger chunks": at most kSHash table-producing gates (one hash
// stage) followed by at most kSXor XOR gates (one XOR stage).  The two halves of a tile run one stage apart,
// so one half always hashes while the other walks its XOR sub-levels.
struct XDesc {           // 8 bytes
    uint32_t lin;        // LDS slot of input 0 | LDS slot of input 1 << 16
    uint32_t lout;       // bits 0-12 LDS slot of the output, bit 13 store-global, bit 14 XNOR, bits 16-31 sub-level id
};
constexpr uint32_t kXStoreGlobal = 1u << 13;
constexpr uint32_t kXXnor = 1u << 14;
constexpr uint32_t kSHash = 256;
constexpr uint32_t kSXor = 768;
struct SChunk {          // 32 bytes
    uint32_t hfirst, n_and, n_or, n_inv;  // hash descriptors [hfirst, hfirst + n_and + n_or + n_inv) of shdescs
    uint32_t xfirst, nx;                  // XOR descriptors [xfirst, xfirst + nx) of sxdescs
    uint32_t pad_[2];
};
static_assert(sizeof(SChunk) == 32, "SChunk must be 32 bytes");

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
    // staggered schedule
    std::vector<FDesc> shdescs;      // hash descriptors (AND, OR, INV order inside a chunk)
    std::vector<uint32_t> shgslot;
    std::vector<XDesc> sxdescs;
    std::vector<uint32_t> sxgslot;
    std::vector<SChunk> schunks;
    std::vector<uint16_t> in_lds;       // LDS slot of every input wire (0xffff: never read)
    uint32_t n_lds_slots = 0;           // high-water mark of live labels
    uint32_t n_hash_phases = 0;
};

// returns GC_OK or GC_E_GATE / GC_E_WIRE / GC_E_ARG
int build_plan(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs, uint32_t noutputs,
               Plan *out);

}  // namespace gc

struct gc_plan {
    gc::Plan p;
};
