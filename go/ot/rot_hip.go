//go:build gchip

package ot

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../mpc_amd/csrc -lgcengine -Wl,-rpath,${SRCDIR}/../../mpc_amd/csrc
#include "gcengine.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// SOURCE ONLY (no Go toolchain in the build image).  Drop-in bodies for the two pad loops of ot/rot.go; everything
// around them — InitSender / InitReceiver, the IKNP calls (already on the device, iknp_hip.go), the seed that travels
// over rot.io (rot.go:139-153, 186-190) — stays the reference's Go.  A maintainer replaces
//
//	rot.go:155-172   pad := make([]Label, 2*otBatchSize); for i := 0; i < len(wires); i += otBatchSize { ... }
//	                 by      return rotSendPads(rot.iknpS.hip.ctx, seed, rot.iknpS.Delta, data, wires)
//	rot.go:193-199   pad := make([]Label, otBatchSize); for i := 0; i < len(flags); i += otBatchSize { ... }
//	                 by      return rotReceivePads(rot.iknpR.hip.ctx, seed, result)
//
// The reference creates a fresh MITCCRH per call (rot.go:143,191), so OT j of a call hashes under key index j; the
// receiver's result[j] then equals the sender's wires[j].L0 / L1 for flag_j = false / true.

// rotSendPads overwrites wires[j] with {H_j(data_j), H_j(data_j ^ delta)} (rot.go:156-172).
func rotSendPads(ctx *C.gc_ctx, seed, delta Label, data []Label, wires []Wire) error {
	if len(data) < len(wires) {
		return fmt.Errorf("ROT: %d IKNP labels for %d wires", len(data), len(wires))
	}
	if len(wires) == 0 {
		return nil
	}
	st := C.gc_rot_send(ctx, (*C.gc_label)(unsafe.Pointer(&seed)), (*C.gc_label)(unsafe.Pointer(&delta)),
		(*C.gc_label)(unsafe.Pointer(&data[0])), C.size_t(len(wires)), (*C.gc_wire)(unsafe.Pointer(&wires[0])))
	if st != C.GC_OK {
		return fmt.Errorf("gcengine: %s: %s", C.GoString(C.gc_strerror(st)), C.GoString(C.gc_last_error()))
	}
	return nil
}

// rotReceivePads hashes result in place (rot.go:194-199): result[j] = H_j(result[j]).
func rotReceivePads(ctx *C.gc_ctx, seed Label, result []Label) error {
	if len(result) == 0 {
		return nil
	}
	st := C.gc_rot_receive(ctx, (*C.gc_label)(unsafe.Pointer(&seed)), (*C.gc_label)(unsafe.Pointer(&result[0])),
		C.size_t(len(result)))
	if st != C.GC_OK {
		return fmt.Errorf("gcengine: %s: %s", C.GoString(C.gc_strerror(st)), C.GoString(C.gc_last_error()))
	}
	return nil
}
