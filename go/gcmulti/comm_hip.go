//go:build gchip

// Package gcmulti — additive multi-GPU driver on top of the circuit shim (SOURCE ONLY: no Go toolchain in the build
// image).  BASELINE config 4: independent instances are sharded over the GPUs of one node (fresh R and labels per
// Garble call, circuit/garble.go:253-278, so nothing is exchanged while garbling / evaluating); the one collective is
// the all-gather of the decoded output bits, ncclAllGather of RCCL over xGMI behind gc_comm_* — no PyTorch.  Every
// device buffer the collective touches comes from the library itself (circuit.DevBuf = gc_dev_alloc): the Go host
// owns no HIP allocator.
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

	"github.com/markkurossi/mpc/circuit"
)

// Node is one process driving the GPUs of the box: one circuit.BatchPipeline (its own gc_ctx, stream and device
// buffers) and one communicator rank per device (gc_comm_init_all = ncclCommInitAll).
type Node struct {
	Pipes []*circuit.BatchPipeline
	comms []*C.gc_comm
	recv  []*circuit.DevBuf // per device: [ranks][perGPU][outputs] decoded bits of every rank
	per   int
	nout  int
}

func statusError(what string, st C.int) error {
	return fmt.Errorf("%s: %s: %s", what, C.GoString(C.gc_strerror(st)), C.GoString(C.gc_last_error()))
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

// Open shards `total` instances of c over the first n devices (n <= 0: all of them): every device gets a pipeline of
// ceil(total / n) instances (the collective needs equal sizes; the last shard is padded), and the ranks join one
// communicator.
func Open(c *circuit.Circuit, total, n int) (*Node, error) {
	if n <= 0 {
		n = int(C.gc_device_count())
	}
	if n < 1 || total < 1 {
		return nil, fmt.Errorf("gcmulti.Open: %d instances on %d devices", total, n)
	}
	node := &Node{per: (total + n - 1) / n, nout: c.Outputs.Size()}
	ctxs := make([]*C.gc_ctx, n)
	for d := 0; d < n; d++ {
		p, err := c.NewBatchPipeline(d, node.per)
		if err != nil {
			node.Close()
			return nil, err
		}
		node.Pipes = append(node.Pipes, p)
		ctxs[d] = (*C.gc_ctx)(p.Ctx())
		r, err := circuit.NewDevBuf(p.Ctx(), n*node.per*node.nout)
		if err != nil {
			node.Close()
			return nil, err
		}
		node.recv = append(node.recv, r)
	}
	node.comms = make([]*C.gc_comm, n)
	if st := C.gc_comm_init_all((**C.gc_ctx)(unsafe.Pointer(&ctxs[0])), C.int(n),
		(**C.gc_comm)(unsafe.Pointer(&node.comms[0]))); st != C.GC_OK {
		node.Close()
		return nil, statusError("gc_comm_init_all", st)
	}
	return node, nil
}

// Run garbles, evaluates and decodes every shard (rnd / bits: the whole batch in instance order, laid out as
// BatchPipeline.SetInputs expects), gathers the decoded bits of all ranks on every device with ONE all-gather behind
// the decode kernels, and returns them in instance order (total x Outputs.Size() bytes) as device 0 holds them.
func (n *Node) Run(key, rnd, bits []byte, total int) ([]byte, error) {
	world := len(n.Pipes)
	nin := len(bits) / total
	rstride := len(rnd) / total
	for d, p := range n.Pipes {
		lo, hi := ShardRange(total, d, world)
		r := make([]byte, n.per*rstride) // the padding instances garble an all-zero stream; their outputs are dropped
		b := make([]byte, n.per*nin)
		copy(r, rnd[lo*rstride:hi*rstride])
		copy(b, bits[lo*nin:hi*nin])
		if err := p.SetInputs(r, b); err != nil {
			return nil, err
		}
	}
	for _, p := range n.Pipes { // asynchronous: all devices run at once
		if err := p.Step(key); err != nil {
			return nil, err
		}
	}
	send := make([]unsafe.Pointer, world)
	recv := make([]unsafe.Pointer, world)
	for d, p := range n.Pipes {
		send[d] = p.OutputsDev().Ptr()
		recv[d] = n.recv[d].Ptr()
	}
	if st := C.gc_comm_allgather_all((**C.gc_comm)(unsafe.Pointer(&n.comms[0])), C.int(world),
		(*unsafe.Pointer)(unsafe.Pointer(&send[0])), (*unsafe.Pointer)(unsafe.Pointer(&recv[0])),
		C.size_t(n.per*n.nout)); st != C.GC_OK {
		return nil, statusError("gc_comm_allgather_all", st)
	}
	all := make([]byte, world*n.per*n.nout)
	if err := n.recv[0].Download(all); err != nil { // waits for device 0's stream: kernels + collective
		return nil, err
	}
	out := make([]byte, 0, total*n.nout)
	for d := 0; d < world; d++ { // drop the padding of the smaller shards (mpc_amd/dist.py: reassemble)
		lo, hi := ShardRange(total, d, world)
		out = append(out, all[d*n.per*n.nout:(d*n.per+hi-lo)*n.nout]...)
	}
	return out, nil
}

// Close releases communicators, buffers and pipelines.
func (n *Node) Close() {
	for i, c := range n.comms {
		if c != nil {
			C.gc_comm_destroy(c)
			n.comms[i] = nil
		}
	}
	for _, r := range n.recv {
		r.Free()
	}
	n.recv = nil
	for _, p := range n.Pipes {
		p.Close()
	}
	n.Pipes = nil
}

// One process PER GPU (the layout of bench.py --gpus N): rank 0 calls UniqueID and hands the 128 bytes to the other
// ranks over the application's own control channel (apps/garbled: its p2p.Conn; bench.py: a file); every rank then
// calls Join with its pipeline's context.
func UniqueID() ([]byte, error) {
	id := make([]byte, C.GC_COMM_ID_BYTES)
	if st := C.gc_comm_get_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0])), C.size_t(len(id))); st != C.GC_OK {
		return nil, statusError("gc_comm_get_unique_id", st)
	}
	return id, nil
}

// Rank is one process's membership of the communicator.
type Rank struct {
	comm *C.gc_comm
	pipe *circuit.BatchPipeline
	recv *circuit.DevBuf
}

// Join creates this rank's communicator on the pipeline's ctx (gc_comm_init_rank; blocks until all nranks joined)
// and the buffer that receives every rank's decoded bits.
func Join(p *circuit.BatchPipeline, id []byte, nranks, rank int) (*Rank, error) {
	var st C.int
	c := C.gc_comm_init_rank((*C.gc_ctx)(p.Ctx()), (*C.uint8_t)(unsafe.Pointer(&id[0])), C.size_t(len(id)),
		C.int(nranks), C.int(rank), &st)
	if c == nil {
		return nil, statusError("gc_comm_init_rank", st)
	}
	recv, err := circuit.NewDevBuf(p.Ctx(), nranks*p.OutputsDev().Size())
	if err != nil {
		C.gc_comm_destroy(c)
		return nil, err
	}
	return &Rank{comm: c, pipe: p, recv: recv}, nil
}

// GatherOutputs enqueues the all-gather of the pipeline's decoded bits behind its last Step and returns every rank's
// bits, rank-major.
func (r *Rank) GatherOutputs() ([]byte, error) {
	n := r.pipe.OutputsDev().Size()
	if st := C.gc_comm_allgather(r.comm, r.pipe.OutputsDev().Ptr(), r.recv.Ptr(), C.size_t(n)); st != C.GC_OK {
		return nil, statusError("gc_comm_allgather", st)
	}
	out := make([]byte, r.recv.Size())
	if err := r.recv.Download(out); err != nil {
		return nil, err
	}
	return out, nil
}

// Close leaves the communicator.
func (r *Rank) Close() {
	if r.comm != nil {
		C.gc_comm_destroy(r.comm)
		r.comm = nil
	}
	r.recv.Free()
}
