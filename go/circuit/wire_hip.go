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
	"unsafe"

	"github.com/markkurossi/mpc/ot"
	"github.com/markkurossi/mpc/p2p"
)

// SOURCE ONLY (no Go toolchain in the build image).  Table egress / ingest of the 2-party driver on the device
// (SURVEY §8f row 1): the tables leave / enter the GPU already in the bytes of the connection, so the host does one
// write / read per circuit instead of the SendUint32 + SendLabel loop over every gate.

// garbleToConn replaces `garbled, err := circ.Garble(rand, key[:])` followed by the "Send garbled tables" loop of
// circuit.Garbler (circuit/garbler.go:53-82).  The returned Garbled carries R and the input / output wire ranges the
// rest of Garbler reads (garbler.go:87,132,153); Gates stays nil — its only reader was the loop this call replaces.
func (c *Circuit) garbleToConn(conn *p2p.Conn, rand io.Reader, key []byte) (*Garbled, error) {
	h, err := c.hip()
	if err != nil {
		return nil, err
	}
	nin, nout := c.Inputs.Size(), c.Outputs.Size()
	rnd := make([]byte, 16*(nin+1))
	if _, err := io.ReadFull(rand, rnd[:16]); err != nil { // R (garble.go:253)
		return nil, err
	}
	if _, err := aes.NewCipher(key); err != nil { // garble.go:260
		return nil, err
	}
	if _, err := io.ReadFull(rand, rnd[16:]); err != nil { // input labels (garble.go:271-278)
		return nil, err
	}
	n := int(C.gc_tables_wire_bytes(h.circ))
	stride := (n + 3) &^ 3
	wire := make([]byte, stride)
	io2 := make([]ot.Wire, nin+nout)
	g := &Garbled{Wires: make([]ot.Wire, c.NumWires)}
	st := C.gc_garble_wire(h.circ, (*C.uint8_t)(unsafe.Pointer(&key[0])), C.size_t(len(key)),
		(*C.uint8_t)(unsafe.Pointer(&rnd[0])), C.size_t(len(rnd)), 1, (*C.gc_label)(unsafe.Pointer(&g.R)),
		(*C.gc_wire)(unsafe.Pointer(&io2[0])), (*C.uint8_t)(unsafe.Pointer(&wire[0])), C.size_t(stride))
	if st != C.GC_OK {
		return nil, statusError(st)
	}
	copy(g.Wires[:nin], io2[:nin])
	copy(g.Wires[c.NumWires-nout:], io2[nin:])
	// BE32(#gates) | per gate BE32(#rows) + rows: the bytes of SendUint32(len(garbled.Gates)) and the loop
	// (garbler.go:69-82), in 64 KiB pieces through the connection's write buffer
	for ofs := 0; ofs < n; {
		if err := conn.NeedSpace(1); err != nil {
			return nil, err
		}
		k := copy(conn.WriteBuf[conn.WritePos:], wire[ofs:n])
		conn.WritePos += k
		ofs += k
		if conn.WritePos == len(conn.WriteBuf) {
			if err := conn.Flush(); err != nil {
				return nil, err
			}
		}
	}
	return g, nil
}

// evalFromConn replaces the "Receive garbled tables" loop of circuit.Evaluator (circuit/evaluator.go:40-66) and the
// later circ.Eval(key, wires, garbled) (evaluator.go:121): the serialised tables are read off the connection in one
// piece and parsed on the device.  wires has the input labels pre-filled, the output range is written.
func (c *Circuit) evalFromConn(conn *p2p.Conn, key []byte, wires []ot.Label) error {
	h, err := c.hip()
	if err != nil {
		return err
	}
	n := int(C.gc_tables_wire_bytes(h.circ))
	stride := (n + 3) &^ 3
	wire := make([]byte, stride)
	ofs := 0
	fill := func(upto int) error { // conn.Fill / ReadBuf are the connection's own read path (p2p/protocol.go:150)
		for ofs < upto {
			if conn.ReadStart == conn.ReadEnd {
				if err := conn.Fill(1); err != nil {
					return err
				}
			}
			k := copy(wire[ofs:upto], conn.ReadBuf[conn.ReadStart:conn.ReadEnd])
			conn.ReadStart += k
			ofs += k
		}
		return nil
	}
	be32 := func(p int) int {
		return int(uint32(wire[p])<<24 | uint32(wire[p+1])<<16 | uint32(wire[p+2])<<8 | uint32(wire[p+3]))
	}
	// The peer's headers are checked as they arrive, like the reference does (evaluator.go:44-47 right after the 4-byte
	// gate count; eval.go:54-56,86-89 for the row counts): a peer that announces another circuit gets the reference's
	// error at once instead of the evaluator blocking on bytes that will never come.
	if err := fill(4); err != nil {
		return err
	}
	if got := be32(0); got != c.NumGates {
		return fmt.Errorf("wrong number of gates: got %d, expected %d", got, c.NumGates)
	}
	for i := range c.Gates {
		if err := fill(ofs + 4); err != nil {
			return err
		}
		want := 0
		switch c.Gates[i].Op {
		case AND:
			want = 2
		case OR:
			want = 3
		case INV:
			want = 1
		}
		if got := be32(ofs - 4); got != want {
			if c.Gates[i].Op == AND {
				return fmt.Errorf("corrupted ciruit: AND row length: %d", got) // eval.go:54-56
			}
			return fmt.Errorf("corrupted circuit: gate %d: %d rows, expected %d", i, got, want) // eval.go:86-89,101-104
		}
		if err := fill(ofs + 16*want); err != nil {
			return err
		}
	}
	nin, nout := c.Inputs.Size(), c.Outputs.Size()
	var bad C.uint32_t
	st := C.gc_eval_wire(h.circ, (*C.uint8_t)(unsafe.Pointer(&key[0])), C.size_t(len(key)), 1,
		(*C.gc_label)(unsafe.Pointer(&wires[0])), (*C.uint8_t)(unsafe.Pointer(&wire[0])), C.size_t(stride),
		(*C.gc_label)(unsafe.Pointer(&wires[c.NumWires-nout])), &bad)
	_ = nin
	if st == C.GC_E_ROWS {
		got := int(uint32(wire[0])<<24 | uint32(wire[1])<<16 | uint32(wire[2])<<8 | uint32(wire[3]))
		if got != c.NumGates {
			return fmt.Errorf("wrong number of gates: got %d, expected %d", got, c.NumGates) // evaluator.go:44-47
		}
		return fmt.Errorf("corrupted ciruit: AND row length") // eval.go:54-56
	}
	if st != C.GC_OK {
		return statusError(st)
	}
	return nil
}
