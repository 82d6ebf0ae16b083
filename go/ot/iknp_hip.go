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

// unpad loop of COT.Receive (cot.go:200-232): result holds the receiver's IKNP labels in, the chosen wire labels out
func cotReceiveUnpad(ctx *C.gc_ctx, seed Label, flags []bool, sent []Label, result []Label) error {
	if len(flags) == 0 {
		return nil
	}
	st := C.gc_cot_receive_unpad(ctx, (*C.gc_label)(unsafe.Pointer(&seed)), (*C.uint8_t)(unsafe.Pointer(&flags[0])),
		(*C.gc_label)(unsafe.Pointer(&sent[0])), (*C.gc_label)(unsafe.Pointer(&result[0])), C.size_t(len(flags)))
	if st != C.GC_OK {
		return fmt.Errorf("gcengine: %s", C.GoString(C.gc_strerror(st)))
	}
	return nil
}

// Send sends n labels (iknp.go:129-194).  The semi-honest part is send(); with malicious set, the KOS consistency
// check — chi-PRG keyed by the receiver's seed2, the unreduced GF(2^128) inner products over result and over the 256
// random choice-vector labels, and the comparison with x * Delta — runs on the device (gc_kos_sender_check); the three
// labels still arrive over s.io exactly as in iknp.go:146-183.
func (s *IKNPSender) Send(n int, malicious bool) ([]Label, error) {
	result, err := s.send(n)
	if err != nil {
		return nil, err
	}
	if !malicious {
		return result, nil
	}
	choiceVector, err := s.send(256)
	if err != nil {
		return nil, err
	}
	var seed2, x, t0, t1 Label
	var ld LabelData
	if err := s.io.ReceiveLabel(&seed2, &ld); err != nil {
		return nil, err
	}
	if err := s.io.ReceiveLabel(&x, &ld); err != nil {
		return nil, err
	}
	if err := s.io.ReceiveLabel(&t0, &ld); err != nil {
		return nil, err
	}
	if err := s.io.ReceiveLabel(&t1, &ld); err != nil {
		return nil, err
	}
	var resPtr *C.gc_label
	if n > 0 {
		resPtr = (*C.gc_label)(unsafe.Pointer(&result[0]))
	}
	var ok C.int
	st := C.gc_kos_sender_check(s.hip.ctx, (*C.gc_label)(unsafe.Pointer(&seed2)), resPtr, C.size_t(n),
		(*C.gc_label)(unsafe.Pointer(&choiceVector[0])), (*C.gc_label)(unsafe.Pointer(&s.Delta)),
		(*C.gc_label)(unsafe.Pointer(&x)), (*C.gc_label)(unsafe.Pointer(&t0)), (*C.gc_label)(unsafe.Pointer(&t1)), &ok)
	if st != C.GC_OK {
		return nil, fmt.Errorf("gcengine: %s", C.GoString(C.gc_strerror(st)))
	}
	if ok == 0 {
		return nil, fmt.Errorf("OT extension check failed") // iknp.go:189-191
	}
	return result, nil
}

// Receive receives labels based on the selection flags b (iknp.go:364-465); the malicious branch draws b0, b1 and
// seed2 from r.rand in the reference's order (:375-382, :404), extends the 256 random choices, and computes
// x, t0, t1 on the device (gc_kos_receiver_tags) before sending them (:454-463).
func (r *IKNPReceiver) Receive(b []bool, result []Label, malicious bool) error {
	if err := r.receive(b, result); err != nil {
		return err
	}
	if !malicious {
		return nil
	}
	b0, err := NewLabel(r.rand)
	if err != nil {
		return err
	}
	b1, err := NewLabel(r.rand)
	if err != nil {
		return err
	}
	bcv := make([]bool, 256)
	for i := 0; i < 256; i++ {
		if i < 128 {
			bcv[i] = b0.Bit(i) == 1
		} else {
			bcv[i] = b1.Bit(i-128) == 1
		}
	}
	choiceVector := make([]Label, 256)
	if err := r.receive(bcv, choiceVector); err != nil {
		return err
	}
	seed2, err := NewLabel(r.rand)
	if err != nil {
		return err
	}
	var ld LabelData
	if err := r.io.SendLabel(seed2, &ld); err != nil {
		return err
	}
	if err := r.io.Flush(); err != nil {
		return err
	}
	var x, t0, t1 Label
	var resPtr *C.gc_label
	var bPtr *C.uint8_t
	if len(b) > 0 {
		resPtr = (*C.gc_label)(unsafe.Pointer(&result[0]))
		bPtr = (*C.uint8_t)(unsafe.Pointer(&b[0]))
	}
	st := C.gc_kos_receiver_tags(r.hip.ctx, (*C.gc_label)(unsafe.Pointer(&seed2)), resPtr, bPtr, C.size_t(len(b)),
		(*C.gc_label)(unsafe.Pointer(&choiceVector[0])), (*C.uint8_t)(unsafe.Pointer(&bcv[0])),
		(*C.gc_label)(unsafe.Pointer(&x)), (*C.gc_label)(unsafe.Pointer(&t0)), (*C.gc_label)(unsafe.Pointer(&t1)))
	if st != C.GC_OK {
		return fmt.Errorf("gcengine: %s", C.GoString(C.gc_strerror(st)))
	}
	if err := r.io.SendLabel(x, &ld); err != nil {
		return err
	}
	if err := r.io.SendLabel(t0, &ld); err != nil {
		return err
	}
	if err := r.io.SendLabel(t1, &ld); err != nil {
		return err
	}
	return r.io.Flush()
}

// SendBits is the bit-COT sender of GMW (iknp.go:259-310): column 0 of the q-matrix, packed little-endian into
// result; the u-matrix chunks arrive framed as in send().
func (s *IKNPSender) SendBits(n int, result []uint64) error {
	if (n+63)/64 > len(result) {
		return fmt.Errorf("result buffer len=%v too short for n=%v", len(result), n)
	}
	want := int(C.gc_iknp_u_bytes(C.size_t(n)))
	u := make([]byte, 0, want)
	for len(u) < want {
		chunk, err := s.io.ReceiveData()
		if err != nil {
			return err
		}
		if len(chunk)%K != 0 {
			return fmt.Errorf("invalid chunk size: %v", len(chunk))
		}
		u = append(u, chunk...)
	}
	if n == 0 {
		return nil
	}
	st := C.gc_iknp_send_bits(s.hip.h, (*C.uint8_t)(unsafe.Pointer(&u[0])), C.size_t(len(u)), C.size_t(n),
		(*C.uint64_t)(unsafe.Pointer(&result[0])))
	if st != C.GC_OK {
		return fmt.Errorf("gcengine: %s", C.GoString(C.gc_strerror(st)))
	}
	return nil
}

// ReceiveBits is the bit-COT receiver of GMW (iknp.go:554-620), including the reference's fold of WHOLE 64-bit
// choice words only (:583-597); the u-matrix leaves in <= 8 KiB messages as in receive().
func (r *IKNPReceiver) ReceiveBits(choices, result []uint64, n int) error {
	if (n+63)/64 > len(choices) {
		return fmt.Errorf("choices buffer len=%v too short for n=%v", len(result), n)
	}
	if (n+63)/64 > len(result) {
		return fmt.Errorf("result buffer len=%v too short for n=%v", len(result), n)
	}
	if n == 0 {
		return r.io.Flush()
	}
	u := make([]byte, int(C.gc_iknp_u_bytes(C.size_t(n))))
	st := C.gc_iknp_receive_bits(r.hip.h, (*C.uint64_t)(unsafe.Pointer(&choices[0])), C.size_t(n),
		(*C.uint8_t)(unsafe.Pointer(&u[0])), (*C.uint64_t)(unsafe.Pointer(&result[0])))
	if st != C.GC_OK {
		return fmt.Errorf("gcengine: %s", C.GoString(C.gc_strerror(st)))
	}
	for ofs := 0; ofs < len(u); ofs += chunkSize {
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
