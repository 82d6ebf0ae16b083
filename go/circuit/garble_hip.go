//go:build gchip

// Package circuit — drop-in bodies for Circuit.Garble / Circuit.Eval on MI355X.
//
// This file is SOURCE ONLY in this repository (the build image has no Go toolchain).  It is what a
// maintainer of markkurossi/mpc adds next to circuit/garble.go and circuit/eval.go: built with
// `-tags gchip` it replaces the two method bodies (the originals get `//go:build !gchip`), every
// exported signature, the Garbled struct and the error values stay as they are, so apps/garbled,
// sha2pc and compiler/ssa compile and run unchanged.
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
	"io"
	"sync"
	"unsafe"

	"github.com/markkurossi/mpc/ot"
)

// HipAllWires makes Garble fill every element of Garbled.Wires (debug / parity runs).  Off by default: the
// reference's callers read only the input range Wires[0:Inputs.Size()) and the output range
// Wires[NumWires-Outputs.Size():] (circuit/garbler.go:87,132,153; sha2pc/garbler.go:115-126), and asking the
// device for all 36 919 wires of aes_128 selects the slower kernel that materialises them.
var HipAllWires = false

// hipCircuit is the device twin of a *Circuit: created on first use, shared by all goroutines
// (gc_circ is immutable; gc_garble / gc_eval are re-entrant).
type hipCircuit struct {
	once sync.Once
	ctx  *C.gc_ctx
	circ *C.gc_circ
	rows int
	err  error
}

var hipCircuits sync.Map // *Circuit -> *hipCircuit

func (c *Circuit) hip() (*hipCircuit, error) {
	v, _ := hipCircuits.LoadOrStore(c, &hipCircuit{})
	h := v.(*hipCircuit)
	h.once.Do(func() {
		var st C.int
		h.ctx = C.gc_ctx_create(0, &st)
		if h.ctx == nil {
			h.err = fmt.Errorf("gcengine: %s: %s", C.GoString(C.gc_strerror(st)), C.GoString(C.gc_last_error()))
			return
		}
		// circuit.Gate is 20 bytes (circuit_test.go:14-19) == gc_gate: the slice crosses cgo as is.
		h.circ = C.gc_circ_load(h.ctx, (*C.gc_gate)(unsafe.Pointer(&c.Gates[0])), C.uint32_t(len(c.Gates)),
			C.uint32_t(c.NumWires), C.uint32_t(c.Inputs.Size()), C.uint32_t(c.Outputs.Size()), &st)
		if h.circ == nil {
			h.err = statusError(st)
			return
		}
		var info C.gc_plan_info
		C.gc_plan_get_info(C.gc_circ_plan(h.circ), &info)
		h.rows = int(info.slab_rows)
	})
	return h, h.err
}

// ReleaseDevice frees the device twin of the circuit (plan, uploaded gate lists, pooled batches) and its context
// (additive).  The reference's Circuit is plain garbage-collected memory; the device copy is not, and the table above keeps
// it alive for as long as the process runs — a server that compiles circuits per request calls this when it drops one.
// Safe to call more than once and for a circuit that never reached the device; a later Garble / Eval loads it again.  Not
// to be called while another goroutine garbles or evaluates the same circuit.
func (c *Circuit) ReleaseDevice() {
	v, ok := hipCircuits.LoadAndDelete(c)
	if !ok {
		return
	}
	h := v.(*hipCircuit)
	h.once.Do(func() {}) // (a concurrent first use has finished, or never starts on this record)
	if h.circ != nil {
		C.gc_circ_free(h.circ)
		h.circ = nil
	}
	if h.ctx != nil {
		C.gc_ctx_destroy(h.ctx)
		h.ctx = nil
	}
}

func statusError(st C.int) error {
	switch st {
	case C.GC_E_KEYSIZE:
		return aes.KeySizeError(0) // callers only test err != nil; text matches "crypto/aes: invalid key size"
	case C.GC_E_GATE:
		return fmt.Errorf("invalid gate type")
	case C.GC_E_ROWS:
		return fmt.Errorf("corrupted ciruit: AND row length")
	default:
		return fmt.Errorf("gcengine: %s: %s", C.GoString(C.gc_strerror(st)), C.GoString(C.gc_last_error()))
	}
}

// Garble garbles the circuit (same contract as circuit/garble.go:248).
func (c *Circuit) Garble(rand io.Reader, key []byte) (*Garbled, error) {
	h, err := c.hip()
	if err != nil {
		return nil, err
	}
	// The io.Reader is consumed exactly like the reference does: R (16 B), then — after the cipher
	// has been created — one 16-byte L0 per input wire (garble.go:253-278).
	nin := c.Inputs.Size()
	rnd := make([]byte, 16*(nin+1))
	if _, err := io.ReadFull(rand, rnd[:16]); err != nil {
		return nil, err
	}
	if _, err := aes.NewCipher(key); err != nil {
		return nil, err
	}
	if _, err := io.ReadFull(rand, rnd[16:]); err != nil {
		return nil, err
	}
	pool := c.garbleScratchPool() // unchanged pool of the reference (garble.go:195-225)
	scratch := pool.Get().(*garbledScratch)
	g := &Garbled{Wires: scratch.wires, Gates: scratch.gates, scratch: scratch, pool: pool}
	var slabPtr *C.gc_label
	if len(scratch.slab) > 0 {
		slabPtr = (*C.gc_label)(unsafe.Pointer(&scratch.slab[0]))
	}
	nout := c.Outputs.Size()
	var wiresPtr, ioPtr *C.gc_wire
	var io []ot.Wire
	if HipAllWires {
		wiresPtr = (*C.gc_wire)(unsafe.Pointer(&scratch.wires[0])) // Garbled.Wires: all wires, both labels
	} else {
		io = make([]ot.Wire, nin+nout)
		ioPtr = (*C.gc_wire)(unsafe.Pointer(&io[0]))
	}
	st := C.gc_garble(h.circ, (*C.uint8_t)(unsafe.Pointer(&key[0])), C.size_t(len(key)),
		(*C.uint8_t)(unsafe.Pointer(&rnd[0])), C.size_t(len(rnd)), 1,
		(*C.gc_label)(unsafe.Pointer(&g.R)), wiresPtr, ioPtr, slabPtr)
	if st != C.GC_OK {
		pool.Put(scratch)
		return nil, statusError(st)
	}
	if !HipAllWires { // the two ranges the callers read; the rest of the pooled slice keeps stale scratch
		copy(scratch.wires[:nin], io[:nin])
		copy(scratch.wires[c.NumWires-nout:], io[nin:])
	}
	// Garbled.Gates[i] = sub-slice of the dense slab (nil for XOR/XNOR), garble.go:290-298
	off := 0
	for i := range c.Gates {
		n := 0
		switch c.Gates[i].Op {
		case AND:
			n = 2
		case OR:
			n = 3
		case INV:
			n = 1
		}
		if n == 0 {
			g.Gates[i] = nil
			continue
		}
		g.Gates[i] = scratch.slab[off : off+n : off+n]
		off += n
	}
	return g, nil
}

// Eval evaluates the circuit (same contract as circuit/eval.go:17).
func (c *Circuit) Eval(key []byte, wires []ot.Label, garbled [][]ot.Label) error {
	h, err := c.hip()
	if err != nil {
		return err
	}
	if _, err := aes.NewCipher(key); err != nil {
		return err
	}
	// flatten [][]ot.Label with the reference's own corruption checks (eval.go:54-56, 86-89, 101-104)
	slab := make([]ot.Label, 0, h.rows)
	for i := range c.Gates {
		row := garbled[i]
		switch c.Gates[i].Op {
		case AND:
			if len(row) != 2 {
				return fmt.Errorf("corrupted ciruit: AND row length: %d", len(row))
			}
		case OR:
			if len(row) < 3 {
				return fmt.Errorf("corrupted circuit: index %d >= row %d", len(row), len(row))
			}
		case INV:
			if len(row) < 1 {
				return fmt.Errorf("corrupted circuit: index %d >= row %d", 0, len(row))
			}
		default:
			continue
		}
		// exactly the rows Eval indexes (row[index-1], eval.go:86-104): a longer row must not shift the next gate's
		switch c.Gates[i].Op {
		case OR:
			row = row[:3]
		case INV:
			row = row[:1]
		}
		slab = append(slab, row...)
	}
	var slabPtr *C.gc_label
	if len(slab) > 0 {
		slabPtr = (*C.gc_label)(unsafe.Pointer(&slab[0]))
	}
	st := C.gc_eval(h.circ, (*C.uint8_t)(unsafe.Pointer(&key[0])), C.size_t(len(key)), 1,
		(*C.gc_label)(unsafe.Pointer(&wires[0])), nil, slabPtr, C.size_t(len(slab)), nil)
	if st != C.GC_OK {
		return statusError(st)
	}
	return nil
}

// GarbleBatch / EvalBatch are ADDITIVE: the reference API is one instance per call; batching is where
// the GPU pays off (see DESIGN.md §7).  rnd holds batch streams of 16*(1+inputs) bytes each.
func (c *Circuit) GarbleBatch(rnd []byte, key []byte, batch int) (R []ot.Label, io []ot.Wire, slab []ot.Label, err error) {
	h, err := c.hip()
	if err != nil {
		return nil, nil, nil, err
	}
	nio := c.Inputs.Size() + c.Outputs.Size()
	R = make([]ot.Label, batch)
	io = make([]ot.Wire, batch*nio)
	slab = make([]ot.Label, batch*h.rows)
	st := C.gc_garble(h.circ, (*C.uint8_t)(unsafe.Pointer(&key[0])), C.size_t(len(key)),
		(*C.uint8_t)(unsafe.Pointer(&rnd[0])), C.size_t(len(rnd)), C.uint32_t(batch),
		(*C.gc_label)(unsafe.Pointer(&R[0])), nil, (*C.gc_wire)(unsafe.Pointer(&io[0])),
		(*C.gc_label)(unsafe.Pointer(&slab[0])))
	if st != C.GC_OK {
		return nil, nil, nil, statusError(st)
	}
	return R, io, slab, nil
}

// EvalBatch is the counterpart of GarbleBatch: inputs holds batch x Inputs.Size() active labels, slab the tables as
// GarbleBatch returned them; the result is batch x Outputs.Size() output labels (Wires[NumWires-Outputs.Size():]).
func (c *Circuit) EvalBatch(key []byte, inputs []ot.Label, slab []ot.Label, batch int) ([]ot.Label, error) {
	h, err := c.hip()
	if err != nil {
		return nil, err
	}
	if len(inputs) != batch*c.Inputs.Size() || len(slab) != batch*h.rows {
		return nil, fmt.Errorf("EvalBatch: %d inputs / %d table labels for batch %d", len(inputs), len(slab), batch)
	}
	out := make([]ot.Label, batch*c.Outputs.Size())
	var slabPtr *C.gc_label
	if len(slab) > 0 {
		slabPtr = (*C.gc_label)(unsafe.Pointer(&slab[0]))
	}
	st := C.gc_eval(h.circ, (*C.uint8_t)(unsafe.Pointer(&key[0])), C.size_t(len(key)), C.uint32_t(batch), nil,
		(*C.gc_label)(unsafe.Pointer(&inputs[0])), slabPtr, C.size_t(h.rows), (*C.gc_label)(unsafe.Pointer(&out[0])))
	if st != C.GC_OK {
		return nil, statusError(st)
	}
	return out, nil
}

