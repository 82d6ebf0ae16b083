//go:build gchip

package circuit

/*
#include "gcengine.h"
*/
import "C"

import (
	"crypto/aes"
	"fmt"
	"unsafe"

	"github.com/markkurossi/mpc/ot"
	"github.com/markkurossi/mpc/p2p"
)

// SOURCE ONLY (no Go toolchain in the build image).  Device twin of circuit.StreamEval
// (circuit/stream_evaluator.go:29-96) and of the per-gate loop of StreamEvaluator's OpCircuit case
// (stream_evaluator.go:270-432).  StreamEvaluator itself (framing, OT, OpResult / OpReturn handling) is unchanged: it
// keeps calling NewStreamEval, Set, SetInputs, Get, and — in place of InitCircuit + the gate loop — evalBlock; with one more
// line at the top of its loop (evalBuffered) whole buffers of blocks go to the device at once.

// StreamEval is a streaming garbled circuit evaluator.
type StreamEval struct {
	ctx *C.gc_ctx
	h   *C.gc_stream_eval
	blk []byte
}

// NewStreamEval creates a new streaming garbled circuit evaluator (stream_evaluator.go:37-50).
func NewStreamEval(key []byte, numInputs, numOutputs int) (*StreamEval, error) {
	if _, err := aes.NewCipher(key); err != nil {
		return nil, err
	}
	var st C.int
	ctx := C.gc_ctx_create(0, &st)
	if ctx == nil {
		return nil, statusError(st)
	}
	h := C.gc_stream_eval_create(ctx, (*C.uint8_t)(unsafe.Pointer(&key[0])), C.size_t(len(key)), &st)
	if h == nil {
		C.gc_ctx_destroy(ctx)
		return nil, statusError(st)
	}
	return &StreamEval{ctx: ctx, h: h}, nil
}

// Get gets the value of a GLOBAL wire (stream_evaluator.go:61-66).  Tmp wires live only inside the OpCircuit block
// that wrote them (stream_garble.go:131-157) and are never read from outside the gate loop.
func (stream *StreamEval) Get(tmp bool, w Wire) ot.Label {
	var l ot.Label
	if tmp || C.gc_stream_eval_get_wire(stream.h, C.uint32_t(w), (*C.gc_label)(unsafe.Pointer(&l))) != C.GC_OK {
		panic(fmt.Sprintf("StreamEval.Get(%v, %d)", tmp, w))
	}
	return l
}

// Set sets the value of a global wire (stream_evaluator.go:76-82).
func (stream *StreamEval) Set(tmp bool, w Wire, label ot.Label) {
	if tmp || C.gc_stream_eval_set_wire(stream.h, C.uint32_t(w), (*C.gc_label)(unsafe.Pointer(&label))) != C.GC_OK {
		panic(fmt.Sprintf("StreamEval.Set(%v, %d)", tmp, w))
	}
}

// SetInputs sets the specified input wire range (stream_evaluator.go:69-73).
func (stream *StreamEval) SetInputs(offset int, inputs []ot.Label) {
	for i := range inputs {
		stream.Set(false, Wire(offset+i), inputs[i])
	}
}

// evalBlock replaces `streaming.InitCircuit(numWires, numTmpWires)` and the `for i := 0; i < numGates; i++` loop
// of the OpCircuit case (stream_evaluator.go:269-432).  The gates' bytes are variable length, so the block is first
// copied off the connection gate by gate (op byte -> sizes; no crypto), then parsed, levelised and evaluated on the
// device in one call.
func (stream *StreamEval) evalBlock(conn *p2p.Conn, numGates, numTmpWires, numWires int) error {
	stream.blk = stream.blk[:0]
	take := func(n int) error { // n bytes from the connection's read buffer (p2p/protocol.go:150)
		for n > 0 {
			if conn.ReadStart == conn.ReadEnd {
				if err := conn.Fill(1); err != nil {
					return err
				}
			}
			k := conn.ReadEnd - conn.ReadStart
			if k > n {
				k = n
			}
			stream.blk = append(stream.blk, conn.ReadBuf[conn.ReadStart:conn.ReadStart+k]...)
			conn.ReadStart += k
			n -= k
		}
		return nil
	}
	for i := 0; i < numGates; i++ {
		if err := take(1); err != nil {
			return err
		}
		gop := stream.blk[len(stream.blk)-1]
		wsz := 4
		if gop&0b00010000 != 0 {
			wsz = 2
		}
		var n int
		switch Operation(gop &^ 0b11110000) {
		case XOR, XNOR:
			n = 3 * wsz
		case AND:
			n = 3*wsz + 2*16
		case OR:
			n = 3*wsz + 3*16
		case INV:
			n = 2*wsz + 16
		default:
			return fmt.Errorf("invalid operation %s", Operation(gop&^0b11110000)) // stream_evaluator.go:340-343
		}
		if err := take(n); err != nil {
			return err
		}
	}
	// also for a block without gates: the call is InitCircuit(numWires, numTmpWires) (stream_evaluator.go:269) — it sizes
	// the wire store, so that a later Get of a wire this block announced does not go out of range
	var consumed C.size_t
	var blkPtr *C.uint8_t
	if len(stream.blk) > 0 {
		blkPtr = (*C.uint8_t)(unsafe.Pointer(&stream.blk[0]))
	}
	st := C.gc_stream_eval_circuit(stream.h, C.uint32_t(numGates), C.uint32_t(numTmpWires), C.uint32_t(numWires),
		blkPtr, C.size_t(len(stream.blk)), &consumed)
	if st != C.GC_OK {
		return statusError(st)
	}
	return nil
}

// evalBuffered is the faster front of the OpCircuit case: called where StreamEvaluator is about to read the next operation
// word (stream_evaluator.go:227), it hands everything the connection has buffered to gc_stream_eval_blocks, which evaluates
// the whole OpCircuit blocks in it (their 20-byte headers included) and says how far it got; the loop then continues with
// whatever operation is next — OpReturn, or an OpCircuit block that did not fit (evalBlock collects that one gate by gate).
// lastStep is the step number of the last block evaluated, for the progress report (:251-268).
// Round 5: a buffer of 64 KiB or more goes to the GPU as it is and the blocks are recognised there (DESIGN.md §5) — the more
// the connection has buffered the better; a ReadBuf from gc_host_alloc (pinnedBytes below) is read by the DMA in place, a
// Go-heap ReadBuf is staged by copier threads of the engine.
func (stream *StreamEval) evalBuffered(conn *p2p.Conn) (blocks int, err error) {
	for {
		have := conn.ReadEnd - conn.ReadStart
		if have < 4 {
			return blocks, nil // (the caller's ReceiveUint32 fills the buffer)
		}
		var consumed C.size_t
		var n C.uint32_t
		var more C.int
		st := C.gc_stream_eval_blocks(stream.h, (*C.uint8_t)(unsafe.Pointer(&conn.ReadBuf[conn.ReadStart])), C.size_t(have),
			&consumed, &n, &more)
		conn.ReadStart += int(consumed)
		blocks += int(n)
		if st != C.GC_OK {
			return blocks, statusError(st)
		}
		if more == 0 {
			return blocks, nil // the next operation is not an OpCircuit block
		}
		if consumed == 0 && have == len(conn.ReadBuf) {
			return blocks, nil // a block larger than the read buffer: evalBlock takes it
		}
		// a block that ends beyond the buffered bytes: move it to the front and read on (p2p/protocol.go:150-170)
		if err := conn.Fill(conn.ReadEnd - conn.ReadStart + 1); err != nil {
			return blocks, err
		}
	}
}

// Stats reports how many OpCircuit blocks were decoded gate by gate and how many were recognised as a block seen before
// up to their rows and global wires (gc_stream_eval_stats; additive, for logging next to the timing lines).
func (stream *StreamEval) Stats() (parsed, matched uint64) {
	var p, m C.uint64_t
	if stream.h != nil {
		C.gc_stream_eval_stats(stream.h, &p, &m)
	}
	return uint64(p), uint64(m)
}

// Close releases the device state (additive).
func (stream *StreamEval) Close() {
	if stream.h != nil {
		C.gc_stream_eval_free(stream.h)
		C.gc_ctx_destroy(stream.ctx)
		stream.h, stream.ctx = nil, nil
	}
}

// pinnedBytes returns n bytes of pinned host memory for a connection's read buffer (p2p.Conn.ReadBuf; additive): whole read
// buffers of OpCircuit blocks then reach the GPU by DMA straight from it (gc_stream_eval_blocks).  The reference sizes its
// buffers at 64 KiB / 1 MiB (p2p/protocol.go:24-25); the engine is happiest with tens of MiB.
func pinnedBytes(n int) ([]byte, func()) {
	p := C.gc_host_alloc(C.size_t(n))
	if p == nil {
		return make([]byte, n), func() {}
	}
	return unsafe.Slice((*byte)(p), n), func() { C.gc_host_free(p) }
}
