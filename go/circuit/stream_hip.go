//go:build gchip

package circuit

/*
#include "gcengine.h"
*/
import "C"

import (
	"crypto/aes"
	"fmt"
	"io"
	"time"
	"unsafe"

	"github.com/markkurossi/mpc/env"
	"github.com/markkurossi/mpc/ot"
	"github.com/markkurossi/mpc/p2p"
)

// SOURCE ONLY (no Go toolchain in the build image).  Drop-in bodies for the streaming garbler,
// circuit/stream_garble.go:41-192: the Streaming type keeps its exported methods (NewStreaming, GetInput,
// GetInputs, Garble) so compiler/ssa.Program.Stream (streamer.go:74-117,568,694) compiles unchanged; the wire
// store and the per-gate loop live behind gc_stream_*.

// Streaming is a streaming garbled circuit garbler (device twin of stream_garble.go:27-38).
type Streaming struct {
	conn    *p2p.Conn
	ctx     *C.gc_ctx
	h       *C.gc_stream
	handles map[*Circuit]streamHandle // gc_stream_intern: a compiled circuit is recognised by content once, not per call
	pending []int               // gate counts of the circuits queued by Begin and not yet written out by Finish
}

// streamHandle names an interned circuit together with the in / out lengths it was interned with (gc_stream_garble_begin_h
// reads exactly that many wire ids from the caller's slices).
type streamHandle struct {
	id        uint32
	nin, nout int
}

// NewStreaming creates a new streaming garbled circuit garbler (stream_garble.go:41-75).  The random stream is
// consumed in the reference's order: R (16 bytes; :46), aes.NewCipher(key) (:52), one L0 per entry of inputs (:67-73).
func NewStreaming(cfg *env.Config, key []byte, inputs []Wire, conn *p2p.Conn) (*Streaming, error) {
	rand := cfg.GetRandom()
	rnd := make([]byte, 16*(len(inputs)+1))
	if _, err := io.ReadFull(rand, rnd[:16]); err != nil {
		return nil, err
	}
	if _, err := aes.NewCipher(key); err != nil {
		return nil, err
	}
	if _, err := io.ReadFull(rand, rnd[16:]); err != nil {
		return nil, err
	}
	var st C.int
	ctx := C.gc_ctx_create(0, &st)
	if ctx == nil {
		return nil, statusError(st)
	}
	var inPtr *C.uint32_t
	if len(inputs) > 0 {
		inPtr = (*C.uint32_t)(unsafe.Pointer(&inputs[0])) // circuit.Wire is uint32 (circuit.go:285)
	}
	h := C.gc_stream_create(ctx, (*C.uint8_t)(unsafe.Pointer(&key[0])), C.size_t(len(key)),
		(*C.uint8_t)(unsafe.Pointer(&rnd[0])), C.size_t(len(rnd)), inPtr, C.uint32_t(len(inputs)), &st)
	if h == nil {
		C.gc_ctx_destroy(ctx)
		return nil, statusError(st)
	}
	return &Streaming{conn: conn, ctx: ctx, h: h}, nil
}

// GetInput gets the value of the input wire (stream_garble.go:117-119).
func (stream *Streaming) GetInput(w Wire) ot.Wire {
	var out ot.Wire
	if C.gc_stream_get_wire(stream.h, C.uint32_t(w), (*C.gc_wire)(unsafe.Pointer(&out))) != C.GC_OK {
		panic(fmt.Sprintf("Streaming.GetInput: wire %d out of range", w)) // the reference indexes out of range
	}
	return out
}

// GetInputs gets the specified input wire range (stream_garble.go:122-128).
func (stream *Streaming) GetInputs(offset, count int) []ot.Wire {
	result := make([]ot.Wire, count)
	for i := 0; i < count; i++ {
		result[i] = stream.GetInput(Wire(offset + i))
	}
	return result
}

// Garble garbles the circuit and streams the garbled tables into the stream (stream_garble.go:161-192): the
// gates' bytes — op|flags, 16/32-bit wire ids, table rows, exactly as garbleGate writes them (:391-446) — come
// back from the device in one piece and go through conn.WriteBuf / NeedSpace like the per-gate loop's output.
// Same contract as the reference: when it returns, the circuit's bytes are in the connection's buffer and its output
// wires are set.  It is Begin followed by Finish; everything queued before is written out first, in order.
func (stream *Streaming) Garble(c *Circuit, in, out []Wire) (time.Duration, time.Duration, error) {
	start := time.Now()
	if err := stream.Begin(c, in, out); err != nil {
		return 0, 0, err
	}
	mid := time.Now()
	for len(stream.pending) > 0 {
		if err := stream.Finish(); err != nil {
			return 0, 0, err
		}
	}
	return mid.Sub(start), time.Since(mid), nil
}

// Begin QUEUES a circuit (additive; gc_stream_garble_begin_h) and returns without waiting for the GPU.  A driver that
// keeps a window of instructions queued ahead — compiler/ssa's streamer loop (streamer.go:412-524) calling Begin for
// instruction k + d before Finish for instruction k — gives the engine step-level parallelism: queued circuits that
// share no wire through in / out are garbled side by side in one launch (include/gcengine.h).  The bytes leave in
// program order through Finish, so the peer sees exactly the stream the reference produces.  in / out must stay
// unchanged until the call returns only (they are copied).
func (stream *Streaming) Begin(c *Circuit, in, out []Wire) error {
	if len(c.Gates) == 0 {
		stream.pending = append(stream.pending, 0)
		return nil
	}
	var inPtr, outPtr *C.uint32_t
	if len(in) > 0 {
		inPtr = (*C.uint32_t)(unsafe.Pointer(&in[0]))
	}
	if len(out) > 0 {
		outPtr = (*C.uint32_t)(unsafe.Pointer(&out[0]))
	}
	h, ok := stream.handles[c]
	if ok && (h.nin != len(in) || h.nout != len(out)) {
		// The handle was interned with the in / out lengths of the circuit's FIRST call and gc_stream_garble_begin_h reads
		// that many ids: a call with other lengths goes by content (the engine then sees another input / output split of the
		// same gate list — as the reference would, which takes the lengths from the slices every time).
		st := C.gc_stream_garble_begin(stream.h, (*C.gc_gate)(unsafe.Pointer(&c.Gates[0])), C.uint32_t(len(c.Gates)),
			C.uint32_t(c.NumWires), inPtr, C.uint32_t(len(in)), outPtr, C.uint32_t(len(out)))
		if st != C.GC_OK {
			return statusError(st)
		}
		stream.pending = append(stream.pending, len(c.Gates))
		return nil
	}
	if !ok {
		var ch C.uint32_t
		st := C.gc_stream_intern(stream.h, (*C.gc_gate)(unsafe.Pointer(&c.Gates[0])), C.uint32_t(len(c.Gates)),
			C.uint32_t(c.NumWires), C.uint32_t(len(in)), C.uint32_t(len(out)), &ch)
		if st != C.GC_OK {
			if st == C.GC_E_GATE {
				return fmt.Errorf("invalid operation") // garbleGate's default case
			}
			return statusError(st)
		}
		if stream.handles == nil {
			stream.handles = make(map[*Circuit]streamHandle)
		}
		h = streamHandle{id: uint32(ch), nin: len(in), nout: len(out)}
		stream.handles[c] = h
	}
	if st := C.gc_stream_garble_begin_h(stream.h, C.uint32_t(h.id), inPtr, outPtr); st != C.GC_OK {
		return statusError(st)
	}
	stream.pending = append(stream.pending, len(c.Gates))
	return nil
}

// Forget gives a circuit's interned device copy back to the engine's bounded cache and lets the *Circuit itself be collected
// (additive; gc_stream_release).  compiler/ssa's streamer keeps one compiled circuit per instruction shape for the whole run
// (streamer.go:467-506), so it never needs this; a host that compiles circuits on the fly and drops them does — without it
// the handles map keeps every Circuit it has ever seen alive and its device copy exempt from eviction.
func (stream *Streaming) Forget(c *Circuit) {
	if h, ok := stream.handles[c]; ok {
		C.gc_stream_release(stream.h, C.uint32_t(h.id))
		delete(stream.handles, c)
	}
}

// Pending is the number of circuits queued by Begin whose bytes Finish has not written out yet.
func (stream *Streaming) Pending() int { return len(stream.pending) }

// Finish writes the bytes of the OLDEST queued circuit into the connection (additive; gc_stream_garble_finish).
func (stream *Streaming) Finish() error {
	if len(stream.pending) == 0 {
		return fmt.Errorf("Streaming.Finish: nothing queued")
	}
	ngates := stream.pending[0]
	stream.pending = stream.pending[1:]
	if ngates == 0 {
		return nil
	}
	// the bytes are read IN PLACE from the engine's pinned staging (valid until the next Finish): the copy into conn.WriteBuf
	// below is the only one
	var ptr *C.uint8_t
	var written C.size_t
	st := C.gc_stream_garble_finish_view(stream.h, &ptr, &written)
	if st != C.GC_OK {
		return statusError(st)
	}
	data := unsafe.Slice((*byte)(unsafe.Pointer(ptr)), int(written))
	for len(data) > 0 { // conn.NeedSpace(512) + direct writes into conn.WriteBuf in the reference (:177-185)
		if err := stream.conn.NeedSpace(512); err != nil {
			return err
		}
		k := copy(stream.conn.WriteBuf[stream.conn.WritePos:], data)
		stream.conn.WritePos += k
		data = data[k:]
	}
	return nil
}

// Close releases the device state (additive: the reference's Streaming is garbage collected).
func (stream *Streaming) Close() {
	if stream.h != nil {
		C.gc_stream_free(stream.h)
		C.gc_ctx_destroy(stream.ctx)
		stream.h, stream.ctx = nil, nil
	}
}
