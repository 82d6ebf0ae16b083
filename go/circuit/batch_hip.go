//go:build gchip

// Device-resident batch pipeline for a Go host (ADDITIVE to the reference API; SOURCE ONLY in this repository: the
// build image has no Go toolchain, tests/test_abi_plan.py checks that every C.gc_* named here is declared in
// include/gcengine.h, and tests/cpp/test_device_pipeline.cpp drives the same calls in the same order from C++ on the
// GPU).
//
// Circuit.Garble / Circuit.Eval (garble_hip.go) keep the reference's one-instance-per-call contract and move every
// table across PCIe.  BatchPipeline is what replaces the loop around them (circuit/garble.go:285-299 /
// circuit/eval.go:37-112 run once per instance) when a caller has MANY independent instances of one circuit
// (BASELINE configs 2 and 4): the random streams go up once, garbled tables, wire labels and the evaluator's state
// stay in HBM, and only the decoded output bits (or the labels a caller asks for) come back.  Every device buffer is
// obtained from the library (gc_dev_alloc / gc_dev_upload / gc_dev_download): the Go side owns no HIP allocator.
package circuit

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../mpc_amd/csrc -lgcengine -Wl,-rpath,${SRCDIR}/../../mpc_amd/csrc
#include "gcengine.h"
*/
import "C"

import (
	"crypto/aes"
	"fmt"
	"unsafe"

	"github.com/markkurossi/mpc/ot"
)

// DevBuf is a device buffer owned through the C ABI.
type DevBuf struct {
	ctx  *C.gc_ctx
	ptr  unsafe.Pointer
	size int
}

// NewDevBuf allocates size bytes on the device of ctx (gc_dev_alloc).  ctx is a gc_ctx* as an unsafe.Pointer: cgo
// types are private to their package, so handles cross package boundaries (gcmulti, ot) untyped.
func NewDevBuf(ctx unsafe.Pointer, size int) (*DevBuf, error) {
	var st C.int
	p := C.gc_dev_alloc((*C.gc_ctx)(ctx), C.size_t(size), &st)
	if p == nil {
		return nil, statusError(st)
	}
	return &DevBuf{ctx: (*C.gc_ctx)(ctx), ptr: p, size: size}, nil
}

// Size is the buffer's length in bytes.
func (d *DevBuf) Size() int { return d.size }

// Ptr is the device address (for gc_comm_allgather and the *_dev OT calls of package ot).
func (d *DevBuf) Ptr() unsafe.Pointer { return d.ptr }

// At is the device address off bytes into the buffer.
func (d *DevBuf) At(off int) unsafe.Pointer { return unsafe.Add(d.ptr, off) }

// Upload copies src to the start of the buffer behind everything queued on the ctx stream; src may be reused on
// return (gc_dev_upload: cgo must not leave Go memory referenced after the call).
func (d *DevBuf) Upload(src []byte) error {
	if len(src) > d.size {
		return fmt.Errorf("DevBuf.Upload: %d bytes into a buffer of %d", len(src), d.size)
	}
	if len(src) == 0 {
		return nil
	}
	if st := C.gc_dev_upload(d.ctx, d.ptr, unsafe.Pointer(&src[0]), C.size_t(len(src))); st != C.GC_OK {
		return statusError(st)
	}
	return nil
}

// Download waits for the ctx stream and copies the first len(dst) bytes of the buffer into dst (gc_dev_download).
func (d *DevBuf) Download(dst []byte) error {
	if len(dst) > d.size {
		return fmt.Errorf("DevBuf.Download: %d bytes from a buffer of %d", len(dst), d.size)
	}
	if len(dst) == 0 {
		return nil
	}
	if st := C.gc_dev_download(d.ctx, unsafe.Pointer(&dst[0]), d.ptr, C.size_t(len(dst))); st != C.GC_OK {
		return statusError(st)
	}
	return nil
}

// Zero fills the buffer with zero bytes (stream-ordered, gc_dev_memset).
func (d *DevBuf) Zero() error {
	if st := C.gc_dev_memset(d.ctx, d.ptr, 0, C.size_t(d.size)); st != C.GC_OK {
		return statusError(st)
	}
	return nil
}

// Free releases the buffer (gc_dev_free waits for the ctx stream first).
func (d *DevBuf) Free() {
	if d != nil && d.ptr != nil {
		C.gc_dev_free(d.ctx, d.ptr)
		d.ptr = nil
	}
}

// BatchPipeline garbles, evaluates and decodes `batch` independent instances of one circuit without leaving HBM.
// Not safe for concurrent use; create one per goroutine (each has its own gc_ctx = its own HIP stream).
type BatchPipeline struct {
	c      *Circuit
	ctx    *C.gc_ctx
	circ   *C.gc_circ
	gb, ev *C.gc_batch
	batch  int
	rows   int
	rnd    *DevBuf // [batch][1+inputs][16]: the bytes each instance's io.Reader would deliver (garble.go:253-278)
	bits   *DevBuf // [batch][inputs] u8: plaintext input bits (both parties'; stands in for "send own labels + OT")
	out    *DevBuf // [batch][outputs] u8: decoded output bits (BitFromLabel, circuit/helpers.go:18-28)
	mis    *DevBuf // u32: output labels that matched neither L0 nor L1
	graph  *C.gc_graph
	key    []byte
}

// NewBatchPipeline loads the circuit on `device` and allocates the state of `batch` instances.
func (c *Circuit) NewBatchPipeline(device, batch int) (*BatchPipeline, error) {
	p := &BatchPipeline{c: c, batch: batch}
	var st C.int
	p.ctx = C.gc_ctx_create(C.int(device), &st)
	if p.ctx == nil {
		return nil, statusError(st)
	}
	p.circ = C.gc_circ_load(p.ctx, (*C.gc_gate)(unsafe.Pointer(&c.Gates[0])), C.uint32_t(len(c.Gates)),
		C.uint32_t(c.NumWires), C.uint32_t(c.Inputs.Size()), C.uint32_t(c.Outputs.Size()), &st)
	if p.circ == nil {
		p.Close()
		return nil, statusError(st)
	}
	var info C.gc_plan_info
	C.gc_plan_get_info(C.gc_circ_plan(p.circ), &info)
	p.rows = int(info.slab_rows)
	if p.gb = C.gc_batch_create(p.circ, C.uint32_t(batch), &st); p.gb == nil {
		p.Close()
		return nil, statusError(st)
	}
	if p.ev = C.gc_batch_create(p.circ, C.uint32_t(batch), &st); p.ev == nil {
		p.Close()
		return nil, statusError(st)
	}
	nin, nout := c.Inputs.Size(), c.Outputs.Size()
	var err error
	ctx := unsafe.Pointer(p.ctx)
	if p.rnd, err = NewDevBuf(ctx, batch*16*(nin+1)); err == nil {
		if p.bits, err = NewDevBuf(ctx, batch*nin); err == nil {
			if p.out, err = NewDevBuf(ctx, batch*nout); err == nil {
				p.mis, err = NewDevBuf(ctx, 4)
			}
		}
	}
	if err != nil {
		p.Close()
		return nil, err
	}
	return p, nil
}

// Ctx is the pipeline's gc_ctx* (for gcmulti and the ot package's *_dev calls on the same stream); untyped because
// cgo types do not cross packages.
func (p *BatchPipeline) Ctx() unsafe.Pointer { return unsafe.Pointer(p.ctx) }

// SetInputs uploads the per-instance random streams — for every instance the bytes Garble's io.Reader would deliver,
// R then one L0 per input wire (garble.go:253-278), 16*(1+Inputs.Size()) bytes — and the plaintext input bits
// (batch x Inputs.Size(), one byte per wire, garbler's inputs first: computer.go:33-39).
func (p *BatchPipeline) SetInputs(rnd []byte, bits []byte) error {
	nin := p.c.Inputs.Size()
	if len(rnd) != p.batch*16*(nin+1) || len(bits) != p.batch*nin {
		return fmt.Errorf("BatchPipeline.SetInputs: %d random bytes / %d bits for batch %d", len(rnd), len(bits), p.batch)
	}
	if err := p.rnd.Upload(rnd); err != nil {
		return err
	}
	return p.bits.Upload(bits)
}

func (p *BatchPipeline) enqueue(key []byte) C.int {
	k := (*C.uint8_t)(unsafe.Pointer(&key[0]))
	if st := C.gc_batch_garble(p.gb, k, C.size_t(len(key)), p.rnd.ptr); st != C.GC_OK {
		return st
	}
	if st := C.gc_batch_select_inputs(p.ev, p.gb, p.bits.ptr); st != C.GC_OK {
		return st
	}
	if st := C.gc_batch_eval(p.ev, k, C.size_t(len(key)), p.gb); st != C.GC_OK {
		return st
	}
	return C.gc_batch_decode(p.gb, p.ev, p.out.ptr, p.mis.ptr)
}

// Step enqueues Garble -> input hand-over -> Eval -> decode for all instances and returns without waiting for the
// GPU.  The first Step with a given key runs the four calls directly (it uploads the round keys) and records them in
// a hipGraph (gc_ctx_capture_*); later Steps with the same key replay that graph with ONE launch.
func (p *BatchPipeline) Step(key []byte) error {
	if _, err := aes.NewCipher(key); err != nil {
		return err // same error as Garble / Eval (garble.go:260, eval.go:20)
	}
	if p.graph != nil && string(p.key) == string(key) {
		if st := C.gc_graph_launch(p.graph); st != C.GC_OK {
			return statusError(st)
		}
		return nil
	}
	if p.graph != nil {
		C.gc_graph_free(p.graph)
		p.graph = nil
	}
	if st := p.enqueue(key); st != C.GC_OK {
		return statusError(st)
	}
	// record the same four calls for the next Step; capture is an optimisation: without it Steps launch directly
	if st := C.gc_ctx_capture_begin(p.ctx); st == C.GC_OK {
		st = p.enqueue(key)
		var g *C.gc_graph
		if st2 := C.gc_ctx_capture_end(p.ctx, &g); st == C.GC_OK && st2 == C.GC_OK {
			p.graph = g
			p.key = append(p.key[:0], key...)
		} else if g != nil {
			C.gc_graph_free(g)
		}
	}
	return nil
}

// Outputs waits for the GPU and returns the decoded output bits of the last Step (batch x Outputs.Size(), one byte
// per output wire) and the number of output labels that were neither L0 nor L1 of their wire (0 unless a table or
// label was corrupted).
func (p *BatchPipeline) Outputs() (bits []byte, mismatches uint32, err error) {
	bits = make([]byte, p.batch*p.c.Outputs.Size())
	if err = p.out.Download(bits); err != nil {
		return nil, 0, err
	}
	var m [4]byte
	if err = p.mis.Download(m[:]); err != nil {
		return nil, 0, err
	}
	return bits, uint32(m[0]) | uint32(m[1])<<8 | uint32(m[2])<<16 | uint32(m[3])<<24, nil
}

// OutputsDev is the device buffer Outputs reads: what gcmulti.AllGather sends to the other ranks (config 4).
func (p *BatchPipeline) OutputsDev() *DevBuf { return p.out }

// Tables returns what the reference's Garbled holds for every instance of the last Step: R, and the dense table slab
// in gate order (Garbled.Gates[i] = slab[row_of_gate[i]:row_of_gate[i+1]], garble.go:290-298) — batch*rows labels.
// This is the PCIe-heavy read-back (238 KB per aes_128 instance); a pipeline that evaluates on the same device never
// needs it.
func (p *BatchPipeline) Tables() (R []ot.Label, slab []ot.Label, err error) {
	R = make([]ot.Label, p.batch)
	slab = make([]ot.Label, p.batch*p.rows)
	if st := C.gc_batch_read_r(p.gb, (*C.gc_label)(unsafe.Pointer(&R[0]))); st != C.GC_OK {
		return nil, nil, statusError(st)
	}
	if len(slab) > 0 {
		if st := C.gc_batch_read_slab(p.gb, (*C.gc_label)(unsafe.Pointer(&slab[0]))); st != C.GC_OK {
			return nil, nil, statusError(st)
		}
	}
	return R, slab, nil
}

// OutputLabels returns the evaluator's active output labels (Wires[NumWires-Outputs.Size():] of every instance).
func (p *BatchPipeline) OutputLabels() ([]ot.Label, error) {
	out := make([]ot.Label, p.batch*p.c.Outputs.Size())
	if st := C.gc_batch_read_outputs(p.ev, (*C.gc_label)(unsafe.Pointer(&out[0]))); st != C.GC_OK {
		return nil, statusError(st)
	}
	return out, nil
}

// LastMs is the HIP-event time of the gate kernels of the last DIRECT (not replayed) garble and eval pass.
func (p *BatchPipeline) LastMs() (garble, eval float32) {
	return float32(C.gc_batch_last_ms(p.gb)), float32(C.gc_batch_last_ms(p.ev))
}

// Sync waits for everything enqueued on the pipeline's stream.
func (p *BatchPipeline) Sync() error {
	if st := C.gc_ctx_sync(p.ctx); st != C.GC_OK {
		return statusError(st)
	}
	return nil
}

// Close releases the device state.
func (p *BatchPipeline) Close() {
	if p.graph != nil {
		C.gc_graph_free(p.graph)
		p.graph = nil
	}
	for _, d := range []*DevBuf{p.rnd, p.bits, p.out, p.mis} {
		d.Free()
	}
	if p.gb != nil {
		C.gc_batch_free(p.gb)
		p.gb = nil
	}
	if p.ev != nil {
		C.gc_batch_free(p.ev)
		p.ev = nil
	}
	if p.circ != nil {
		C.gc_circ_free(p.circ)
		p.circ = nil
	}
	if p.ctx != nil {
		C.gc_ctx_destroy(p.ctx)
		p.ctx = nil
	}
}
