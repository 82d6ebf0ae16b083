//go:build gchip

// Package ot — drop-in bodies for the IKNP expansion loops and the COT pad loops on MI355X.
// SOURCE ONLY here (no Go toolchain in the build image); see INTEGRATION.md.
//
// The exported API of ot/iknp.go and ot/cot.go is untouched: NewIKNPSender / NewIKNPReceiver still run
// the 128 base OTs in Go (EC P-256 stays on the CPU) and frame the u-matrix as <= 8 KiB messages on
// ot.IO exactly like iknp.go:499 / :203, so a GPU party interoperates with an unmodified Go party.
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

// hipIKNP replaces the [K]cipher.Stream arrays g0 / g1 of IKNPSender / IKNPReceiver (iknp.go:80-86,
// 313-318): the per-column AES-128-CTR streams and their byte position live on the device.
type hipIKNP struct {
	ctx *C.gc_ctx
	h   *C.gc_iknp
}

// after base.Send(wires[:]) in NewIKNPReceiver (iknp.go:337-356)
func newHipReceiver(ctx *C.gc_ctx, wires *[K]Wire) (*hipIKNP, error) {
	var st C.int
	h := C.gc_iknp_receiver_create(ctx, (*C.gc_wire)(unsafe.Pointer(&wires[0])), &st)
	if h == nil {
		return nil, fmt.Errorf("gcengine: %s", C.GoString(C.gc_strerror(st)))
	}
	return &hipIKNP{ctx: ctx, h: h}, nil
}

// after base.Receive(flags[:], k0[:]) in NewIKNPSender (iknp.go:112-122)
func newHipSender(ctx *C.gc_ctx, delta Label, k0 *[K]Label) (*hipIKNP, error) {
	var st C.int
	h := C.gc_iknp_sender_create(ctx, (*C.gc_label)(unsafe.Pointer(&delta)),
		(*C.gc_label)(unsafe.Pointer(&k0[0])), &st)
	if h == nil {
		return nil, fmt.Errorf("gcengine: %s", C.GoString(C.gc_strerror(st)))
	}
	return &hipIKNP{ctx: ctx, h: h}, nil
}

// body of (*IKNPReceiver).receive (iknp.go:468-511)
func (r *IKNPReceiver) receive(b []bool, result []Label) error {
	if len(b) != len(result) {
		panic("len(b) != len(result)")
	}
	n := len(b)
	if n == 0 {
		return r.io.Flush()
	}
	u := make([]byte, int(C.gc_iknp_u_bytes(C.size_t(n))))
	// []bool is one byte per element (0/1): it crosses cgo as the choice array
	st := C.gc_iknp_receive(r.hip.h, (*C.uint8_t)(unsafe.Pointer(&b[0])), C.size_t(n),
		(*C.uint8_t)(unsafe.Pointer(&u[0])), (*C.gc_label)(unsafe.Pointer(&result[0])))
	if st != C.GC_OK {
		return fmt.Errorf("gcengine: %s", C.GoString(C.gc_strerror(st)))
	}
	for ofs := 0; ofs < len(u); ofs += chunkSize { // same framing as iknp.go:499
		end := ofs + chunkSize
		if end > len(u) {
			end = len(u)
		}
		if err := r.io.SendData(u[ofs:end]); err != nil {
			return err
		}
	}
	return r.io.Flush()
}

// body of (*IKNPSender).send (iknp.go:197-226)
func (s *IKNPSender) send(n int) ([]Label, error) {
	result := make([]Label, n)
	want := int(C.gc_iknp_u_bytes(C.size_t(n)))
	u := make([]byte, 0, want)
	for len(u) < want {
		chunk, err := s.io.ReceiveData()
		if err != nil {
			return nil, err
		}
		if len(chunk)%K != 0 {
			return nil, fmt.Errorf("invalid chunk size: %v", len(chunk))
		}
		u = append(u, chunk...)
	}
	if n == 0 {
		return result, nil
	}
	st := C.gc_iknp_send(s.hip.h, (*C.uint8_t)(unsafe.Pointer(&u[0])), C.size_t(len(u)), C.size_t(n),
		(*C.gc_label)(unsafe.Pointer(&result[0])))
	if st != C.GC_OK {
		return nil, fmt.Errorf("invalid chunk size: %v", len(u))
	}
	return result, nil
}

// pad loop of COT.Send (cot.go:155-182): out = the 2n labels that go on the wire, in order
func cotSendPads(ctx *C.gc_ctx, seed, delta Label, data []Label, wires []Wire) ([]Label, error) {
	out := make([]Label, 2*len(wires))
	if len(wires) == 0 {
		return out, nil
	}
	st := C.gc_cot_send_pads(ctx, (*C.gc_label)(unsafe.Pointer(&seed)), (*C.gc_label)(unsafe.Pointer(&delta)),
		(*C.gc_label)(unsafe.Pointer(&data[0])), (*C.gc_wire)(unsafe.Pointer(&wires[0])), C.size_t(len(wires)),
		(*C.gc_label)(unsafe.Pointer(&out[0])))
	if st != C.GC_OK {
		return nil, fmt.Errorf("gcengine: %s", C.GoString(C.gc_strerror(st)))
	}
	return out, nil
}

// (*MITCCRH).Hash over a whole run (mitccrh.go:93-128 as driven by cot.go:160-171)
func mitccrhHashAll(ctx *C.gc_ctx, seed Label, gid0 uint64, blks []Label, h int) error {
	if len(blks) == 0 {
		return nil
	}
	st := C.gc_mitccrh_hash(ctx, (*C.gc_label)(unsafe.Pointer(&seed)), C.uint64_t(gid0),
		(*C.gc_label)(unsafe.Pointer(&blks[0])), C.size_t(len(blks)/h), C.uint32_t(h))
	if st != C.GC_OK {
		return fmt.Errorf("gcengine: %s", C.GoString(C.gc_strerror(st)))
	}
	return nil
}
