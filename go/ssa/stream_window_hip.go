//go:build gchip

// Package ssa — a window of queued instructions for Program.Stream on MI355X.
//
// SOURCE ONLY in this repository (no Go toolchain in the build image), like the rest of go/.  The reference garbles one
// instruction at a time: Program.garble (compiler/ssa/streamer.go:664-704) writes the OpCircuit header into the connection and
// calls (*circuit.Streaming).Garble, which returns when the instruction's bytes are in conn.WriteBuf.  That contract still
// holds with the engine underneath (Garble = Begin + Finish), but a GPU that sees one small circuit at a time is a launch
// sequence per instruction (~35 us: below the CPU for a 64-bit adder).  What the engine is built for is a WINDOW: Begin for
// instruction k + depth before Finish for instruction k.  Queued circuits that share no wire are garbled side by side in one
// launch, long ones run on lanes of their own, and the bytes still leave in program order — header k, then bytes k — so the
// peer sees exactly the stream the reference produces (DESIGN.md §5; 4 x 10^8 gates/s on the Ed25519 inner loop with 1 024
// instructions queued ahead, against 1.4 x 10^7 published for the reference).
//
// The patch to streamer.go is three edits:
//
//	(1) in Program.Stream, next to `streaming, err := circuit.NewStreaming(...)` (streamer.go:74):
//	        win := newGarbleWindow(conn, streaming, 1024)
//	(2) Program.garble (streamer.go:664-704) becomes
//	        func (prog *Program) garble(win *garbleWindow, step int, circ *circuit.Circuit, in, out []circuit.Wire) error {
//	                prog.stats.Add(circ.Stats)
//	                prog.numWires += circ.NumWires
//	                return win.push(step, circ, in, out)
//	        }
//	(3) win.drain() wherever the streamer writes to the connection itself or reads a wire back: in front of
//	    conn.SendUint32(circuit.OpReturn) in the Ret case (streamer.go:393) and of streaming.GetInput / GetInputs.
//
// Nothing else changes: the wire format, the order of everything on the connection and StreamEvaluator stay as they are.
package ssa

import (
	"github.com/markkurossi/mpc/circuit"
	"github.com/markkurossi/mpc/p2p"
)

// queuedStep is the OpCircuit header of an instruction whose circuit the engine has queued (streamer.go:679-693).
type queuedStep struct {
	step, numGates, numWires, maxID int
}

// garbleWindow keeps up to depth instructions queued in the engine ahead of the one whose bytes are being written.
type garbleWindow struct {
	conn      *p2p.Conn
	streaming *circuit.Streaming
	depth     int
	queued    []queuedStep
}

func newGarbleWindow(conn *p2p.Conn, streaming *circuit.Streaming, depth int) *garbleWindow {
	if depth < 1 {
		depth = 1
	}
	return &garbleWindow{conn: conn, streaming: streaming, depth: depth}
}

// push queues one instruction (Streaming.Begin: no wait for the GPU) and, when the window is full, writes out the oldest.
func (w *garbleWindow) push(step int, circ *circuit.Circuit, in, out []circuit.Wire) error {
	var maxID circuit.Wire
	for _, id := range in {
		if id > maxID {
			maxID = id
		}
	}
	for _, id := range out {
		if id > maxID {
			maxID = id
		}
	}
	if err := w.streaming.Begin(circ, in, out); err != nil {
		return err
	}
	w.queued = append(w.queued, queuedStep{step: step, numGates: circ.NumGates, numWires: circ.NumWires, maxID: int(maxID)})
	if len(w.queued) > w.depth {
		return w.pop()
	}
	return nil
}

// pop writes the oldest queued instruction to the connection: its header exactly as Program.garble writes it
// (streamer.go:679-693), then its gates (Streaming.Finish: the reference's bytes, stream_garble.go:391-446).
func (w *garbleWindow) pop() error {
	q := w.queued[0]
	w.queued = w.queued[1:]
	for _, v := range []int{circuit.OpCircuit, q.step, q.numGates, q.numWires, q.maxID + 1} {
		if err := w.conn.SendUint32(v); err != nil {
			return err
		}
	}
	return w.streaming.Finish()
}

// drain writes out everything queued (before the streamer writes to the connection itself or reads a wire back).
func (w *garbleWindow) drain() error {
	for len(w.queued) > 0 {
		if err := w.pop(); err != nil {
			return err
		}
	}
	return nil
}
