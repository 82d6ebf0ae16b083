//go:build gchip

package circuit

/*
#include "gcengine.h"
*/
import "C"

import (
	"os"
	"unsafe"

	"github.com/markkurossi/mpc/ot"
)

// A streaming host wants eight hardware queues: the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES of
// them (4 by default) and streams that share one run one after the other (mpc_amd/csrc/engine.cpp).  The runtime reads the
// variable on the first HIP call of the process, so it is set here, before any — by the HOST: libgcengine.so does not touch
// the environment.  A value the operator has set is kept.
func init() {
	if _, ok := os.LookupEnv("GPU_MAX_HW_QUEUES"); !ok {
		os.Setenv("GPU_MAX_HW_QUEUES", "8")
	}
}

// SOURCE ONLY (no Go toolchain in the build image).  The reference pools its garble scratch on the Go heap
// (garbleScratchPool, circuit/garble.go:195-225).  Backing the slab of that pool with pinned memory (gc_host_alloc =
// hipHostMalloc) lets gc_garble / gc_eval DMA the tables straight into / out of it and pipeline the copy against the
// layout transposes (include/gcengine.h, "Pinned host memory").  Used by newGarbledScratch in place of make().

// pinnedLabels returns a []ot.Label of n elements in pinned host memory and the function that frees it.
func pinnedLabels(n int) ([]ot.Label, func()) {
	if n == 0 {
		return nil, func() {}
	}
	p := C.gc_host_alloc(C.size_t(n) * C.size_t(unsafe.Sizeof(ot.Label{})))
	if p == nil {
		return make([]ot.Label, n), func() {} // pageable fall-back: same results, runtime-staged copies
	}
	return unsafe.Slice((*ot.Label)(p), n), func() { C.gc_host_free(p) }
}
