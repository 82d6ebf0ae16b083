// Host-side byte loops of the streaming evaluator's skeleton match (stream_engine.cpp, EvalSkel), in a translation unit
// of their own because they are x86 SIMD code the device pass of a HIP source cannot see.
//
// A block the evaluator has met before is compared as ONE masked run — (block ^ reference) & mask over every byte, the mask
// zero under the global ids and the table rows — instead of one short compare per gate with rows: a 4 000-gate multiplier
// block is 1 200 runs of ~20 bytes, and the run bookkeeping, not the bytes, was the cost (tools/skel_match_ubench.cpp:
// 14.0 -> 9.1 us per 66 KB block on one core; the block comes from DRAM in 6.7 us).  Rows are moved with one byte shuffle
// each: wire order BE(D0) || BE(D1) (label.go:105-108) -> {D0, D1} in host order.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/gcengine.h"

namespace skel_simd {
// 2: AVX2 compare + SSSE3 rows, 1: SSE2 compare + SSSE3 rows, 0: SSE2 compare + plain rows (GC_STREAM_PLAIN_MATCH forces 0)
int level();
// (buf ^ ref) & mask == 0 over n bytes
bool same(const uint8_t *buf, const uint8_t *ref, const uint8_t *mask, size_t n);
// dst[r] = the 16-byte row at buf + off[r], both halves byte-swapped
void rows(const uint8_t *buf, const uint32_t *off, size_t n, gc_label *dst);
}  // namespace skel_simd
