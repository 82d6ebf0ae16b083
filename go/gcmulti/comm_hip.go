//go:build gchip

// Package gcmulti — additive multi-GPU driver on top of the circuit shim (SOURCE ONLY: no Go toolchain in the build
// image).  BASELINE config 4: independent instances are sharded over the GPUs of one node (fresh R and labels per
// Garble call, circuit/garble.go:253-278, so nothing is exchanged while garbling / evaluating); the one collective is
// the all-gather of the decoded output bits, ncclAllGather of RCCL over xGMI behind gc_comm_* — no PyTorch.
package gcmulti

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

// Node is one process driving every GPU of the box: one gc_ctx and one communicator rank per device
// (gc_comm_init_all = ncclCommInitAll).
type Node struct {
	Ctxs  []*C.gc_ctx
	Comms []*C.gc_comm
}

func statusError(what string, st C.int) error {
	return fmt.Errorf("%s: %s: %s", what, C.GoString(C.gc_strerror(st)), C.GoString(C.gc_last_error()))
}

// Open creates the contexts and the communicator over the first n devices (n <= 0: all of them).
func Open(n int) (*Node, error) {
	if n <= 0 {
		n = int(C.gc_device_count())
	}
	node := &Node{Ctxs: make([]*C.gc_ctx, n), Comms: make([]*C.gc_comm, n)}
	for d := 0; d < n; d++ {
		var st C.int
		node.Ctxs[d] = C.gc_ctx_create(C.int(d), &st)
		if node.Ctxs[d] == nil {
			node.Close()
			return nil, statusError("gc_ctx_create", st)
		}
	}
	if st := C.gc_comm_init_all((**C.gc_ctx)(unsafe.Pointer(&node.Ctxs[0])), C.int(n),
		(**C.gc_comm)(unsafe.Pointer(&node.Comms[0]))); st != C.GC_OK {
		node.Close()
		return nil, statusError("gc_comm_init_all", st)
	}
	return node, nil
}

// ShardRange is the contiguous instance range of rank when total instances are split over world ranks
// (mpc_amd/dist.py: shard_range).
func ShardRange(total, rank, world int) (lo, hi int) {
	base, rem := total/world, total%world
	lo = rank*base + min(rank, rem)
	hi = lo + base
	if rank < rem {
		hi++
	}
	return lo, hi
}

// AllGather gathers bytes per rank from the device buffer send[d] of every rank into recv[d] (len(Comms)*bytes,
// rank-major) on every device: the terminal exchange of a sharded batch (decoded bits of gc_batch_decode, or the
// output labels of gc_batch_gather_outputs).  Asynchronous on the ranks' ctx streams; Sync waits for them.
func (n *Node) AllGather(send, recv []unsafe.Pointer, bytes int) error {
	if st := C.gc_comm_allgather_all((**C.gc_comm)(unsafe.Pointer(&n.Comms[0])), C.int(len(n.Comms)),
		(*unsafe.Pointer)(unsafe.Pointer(&send[0])), (*unsafe.Pointer)(unsafe.Pointer(&recv[0])), C.size_t(bytes)); st != C.GC_OK {
		return statusError("gc_comm_allgather_all", st)
	}
	return nil
}

// Sync waits for everything enqueued on every rank's stream.
func (n *Node) Sync() error {
	for _, c := range n.Ctxs {
		if st := C.gc_ctx_sync(c); st != C.GC_OK {
			return statusError("gc_ctx_sync", st)
		}
	}
	return nil
}

// Close releases communicators and contexts.
func (n *Node) Close() {
	for i, c := range n.Comms {
		if c != nil {
			C.gc_comm_destroy(c)
			n.Comms[i] = nil
		}
	}
	for i, c := range n.Ctxs {
		if c != nil {
			C.gc_ctx_destroy(c)
			n.Ctxs[i] = nil
		}
	}
}

// One process PER GPU (the layout of bench.py --gpus N): rank 0 calls UniqueID and hands the 128 bytes to the other
// ranks over the application's own control channel (apps/garbled: its p2p.Conn); every rank then calls Join.
func UniqueID() ([]byte, error) {
	id := make([]byte, C.GC_COMM_ID_BYTES)
	if st := C.gc_comm_get_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0])), C.size_t(len(id))); st != C.GC_OK {
		return nil, statusError("gc_comm_get_unique_id", st)
	}
	return id, nil
}

// Join creates this rank's communicator on ctx (gc_comm_init_rank; blocks until all nranks have joined).
func Join(ctx *C.gc_ctx, id []byte, nranks, rank int) (*C.gc_comm, error) {
	var st C.int
	c := C.gc_comm_init_rank(ctx, (*C.uint8_t)(unsafe.Pointer(&id[0])), C.size_t(len(id)), C.int(nranks), C.int(rank), &st)
	if c == nil {
		return nil, statusError("gc_comm_init_rank", st)
	}
	return c, nil
}
