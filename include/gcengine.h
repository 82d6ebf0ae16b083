/*
 * gcengine.h — C ABI of the MI355X garbled-circuit engine (libgcengine.so).
 *
 * This is the drop-in boundary for the hot path of markkurossi/mpc.  The reference is pure
 * Go and has no FFI layer; each entry point below names the exported Go function whose BODY
 * it replaces (the cgo stub a maintainer adds is shown in INTEGRATION.md and go/).
 * Plain C: pointers + sizes only, no callbacks, every function re-entrant.  All pointers
 * are caller-owned and only used for the duration of the call unless stated otherwise.
 *
 * Memory layouts are Go's in-memory structs, so Go slices cross cgo without copies:
 *   gc_label == ot.Label   {D0,D1 uint64}            ot/label.go:28-31
 *   gc_wire  == ot.Wire    {L0,L1 Label}             ot/label.go:18-21
 *   gc_gate  == circuit.Gate (20 bytes)              circuit/circuit.go:260-266
 */
#ifndef GCENGINE_H
#define GCENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gc_label { uint64_t d0, d1; } gc_label;
typedef struct gc_wire { gc_label l0, l1; } gc_wire;
typedef struct gc_gate {
    uint32_t in0, in1, out;
    uint8_t op; /* circuit.Operation: XOR=0 XNOR=1 AND=2 OR=3 INV=4 (circuit.go:25-34) */
    uint8_t pad_[3];
    uint32_t level;
} gc_gate;

enum {
    GC_XOR = 0, GC_XNOR = 1, GC_AND = 2, GC_OR = 3, GC_INV = 4
};

/* status codes; the Go shim maps them to the reference's error values */
enum {
    GC_OK = 0,
    GC_E_KEYSIZE = -1, /* aes.NewCipher: "crypto/aes: invalid key size" (garble.go:260, eval.go:20) */
    GC_E_RAND = -2,    /* random stream shorter than R + inputs (io.Reader error, garble.go:253,272) */
    GC_E_GATE = -3,    /* "invalid gate type" / "invalid operation" (garble.go:326, eval.go:43) */
    GC_E_ROWS = -4,    /* "corrupted circuit": table shorter than the gates need (eval.go:54-56,86-89) */
    GC_E_ARG = -5,     /* NULL / size mismatch */
    GC_E_HIP = -6,     /* HIP runtime failure (no device, launch error); see gc_last_error() */
    GC_E_NOMEM = -7,   /* device or host allocation failed */
    GC_E_WIRE = -8     /* gate reads a wire that no input/gate has written, or wire id >= nwires */
};

const char *gc_strerror(int status);
/* thread-local detail of the last GC_E_HIP / GC_E_NOMEM on this thread ("" if none) */
const char *gc_last_error(void);
/* ABI version of this header (checked by the bindings).  2: + gc_dev_* (device memory for hosts without a HIP
 * allocator), gc_rot_* (ot/rot.go), gc_stream_garble_flush / gc_stream_stats / gc_stream_intern / gc_stream_garble_begin_h and up
 * to 4 096 circuits in flight per stream (step groups); every v1 entry point is unchanged. */
#define GC_ABI_VERSION 2
int gc_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Circuit plan — host-only "compile" of a gate list (no GPU needed).
 * Replaces nothing in the reference; it is the levelised re-encoding the device needs
 * (reference analogue: Circuit.AssignLevels, circuit/circuit.go:206-254, TargetYao).
 * Gates are renamed to single-assignment slots, bucketed by dependency level, and given the
 * tweak / table-row prefix sums of the reference's serial loop (garble.go:357-359,419-420,
 * 451-452 and :199-211) so that level-parallel execution is bit-identical to it.
 * ------------------------------------------------------------------------------------------ */
typedef struct gc_plan gc_plan;

typedef struct gc_plan_info {
    uint32_t ngates, nwires, ninputs, noutputs;
    uint32_t nlevels;   /* == Stats[NumLevels] of AssignLevels(TargetYao) */
    uint32_t max_width; /* == Stats[MaxWidth] */
    uint32_t slab_rows; /* table labels per instance: AND 2, OR 3, INV 1 */
    uint32_t n_xor, n_xnor, n_and, n_or, n_inv;
    uint32_t nslots;    /* device wire slots = ninputs + ngates */
    uint32_t n_steps;   /* dependency levels = launches per pass of schedule 0 */
    uint32_t n_hash_phases; /* fused schedule: steps that hash (non-free depth of the circuit) */
    uint32_t n_fused_steps; /* fused schedule: hash phases + XOR sub-levels */
    uint32_t n_lds_slots;   /* fused schedule: peak number of live wire labels (0xffffffff: > 65534) */
    /* flattened fused schedule: every NEEDED XOR output is one XOR over a list of earlier labels */
    uint32_t n_flat_slots;  /* peak live labels + the zero slot (0xffffffff: not available) */
    uint32_t n_flat_outs;   /* XOR outputs that are materialised (of n_xor + n_xnor gates) */
    uint32_t n_flat_terms;  /* labels read by them in total */
    uint32_t n_flat_steps;  /* hash phases + XOR rounds = workgroup barriers per pass */
    uint32_t n_flat_units;  /* staging units */
} gc_plan_info;

gc_plan *gc_plan_create(const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                        uint32_t noutputs, int *status);
/* The plan a CHAIN of streamed circuits runs on when the streaming engine fuses them into one job (round 5, chain fusion:
 * gc_stream_fuse_stats below; mpc_amd/csrc/stream_fuse.cpp).  Step k has gates[k][0 .. ngates[k]) in its own wire ids —
 * inputs [0, nin[k]), outputs the last nout[k] wires — and wiring[k][i] says where its input i comes from: 0xffffffff = the
 * stream's wire store, else (m << 24 | j) = output j of the earlier step m (wiring[0] may be NULL: all from the store).  The
 * merged circuit reads the store inputs in step order, its outputs are every step's outputs in step order, the hash tweak
 * starts over at every step (circuit/stream_garble.go:174) and the table rows count through.  Host only; the introspection
 * calls below apply (gc_plan_describe: gate k of step s is gate first_gate_of_step[s] + k of the merged list). */
gc_plan *gc_plan_create_chain(const gc_gate *const *gates, const uint32_t *ngates, const uint32_t *nwires, const uint32_t *nin,
                              const uint32_t *nout, const uint32_t *const *wiring, uint32_t nsteps, int *status);
void gc_plan_free(gc_plan *);
int gc_plan_get_info(const gc_plan *, gc_plan_info *out);
/* introspection used by the parity tests (arrays sized by the caller):
 *  level_of_gate[ngates]  reference gate.Level for every gate (original order)
 *  tweak_of_gate[ngates]  value of `id` when the serial loop reaches the gate
 *  row_of_gate[ngates+1]  first slab row of the gate (== Garbled.Gates[i] offset into the slab)
 *  slot_of_gate[ngates]   device wire slot written by the gate
 * any pointer may be NULL */
/* Host-side self-check of the flattened fused schedule (no GPU involved): evaluates the circuit on plaintext bits by
 * walking the exact unit program, LDS slot assignment, part joins and global stores the fused kernels execute, with
 * their parallel semantics.  in_bits[ninputs], out_bits[noutputs] (one byte per bit).  GC_E_WIRE: a slot was read
 * before it was written; GC_E_ARG: no flattened schedule / malformed items. */
int gc_plan_simulate(const gc_plan *, const uint8_t *in_bits, uint8_t *out_bits);
int gc_plan_describe(const gc_plan *, uint32_t *level_of_gate, uint32_t *tweak_of_gate,
                     uint32_t *row_of_gate, uint32_t *slot_of_gate);
/* 64-bit fingerprint of the device program of the plan (level steps, hash-phase schedule, flattened unit program and LDS
 * slots): equal fingerprints = the same work launched for this circuit.  bench.py ties PMC counters to it (round 6). */
int gc_plan_fingerprint(const gc_plan *, uint64_t *fp);

/* ------------------------------------------------------------------------------------------
 * Device context + circuit
 * ------------------------------------------------------------------------------------------ */
typedef struct gc_ctx gc_ctx;   /* one HIP device + one stream; create one per goroutine/thread for concurrency */
typedef struct gc_circ gc_circ; /* a plan uploaded to a device; immutable, shareable between threads */

int gc_device_count(void);
gc_ctx *gc_ctx_create(int device, int *status);
void gc_ctx_destroy(gc_ctx *);
int gc_ctx_sync(gc_ctx *);
/* ONE instance of a wide circuit (a big streamed step) runs as one launch of 32 workgroups that meet at a barrier between
 * levels; their wait is bounded, and a pass that lost a workgroup is done again on the device, on the same stream, before
 * anything that follows it (so no call fails and no result differs: (*Streaming).Garble of the reference never fails
 * spuriously, circuit/stream_garble.go:161-192) — it costs time, and the ctx keeps to one launch per level from then on.
 * state: 0 not used yet, 1 in use, -1 off (GC_NO_COOP, the self-test failed, or after a timeout); timeouts: passes that were
 * done again.  Either pointer may be NULL.  (GC_COOP_FORCE_TIMEOUT=n makes the n-th pass of every ctx lose a workgroup.) */
int gc_ctx_coop_stats(gc_ctx *, int *state, uint64_t *timeouts);
/* which device the ctx really sits on: its PCI bus id as text ("0000:75:00.0", hipDeviceGetPCIBusId) — a multi-GPU launcher
 * that hands every rank its own HIP_VISIBLE_DEVICES shows each process ONE device, number 0; the bus id tells the ranks'
 * devices apart (bench.py --gpus N records it per rank in its config4 block).  len >= 16. */
int gc_ctx_pci_bus_id(gc_ctx *, char *buf, size_t len);
/* the ctx's HIP stream as an opaque pointer (hipStream_t) for callers that enqueue their own work */
void *gc_ctx_stream(gc_ctx *);

/* Pipeline graphs (hipGraph): record a sequence of DEVICE-RESIDENT calls on this ctx (gc_batch_garble,
 * _select_inputs, _eval, _decode, _egress_tables, gc_cot_*_dev, ...; same pointers every time) once, then replay it
 * with a single launch — removes the host launch latency between the five-odd kernels of a step.
 * Between begin and end nothing executes; host-buffer calls (gc_garble, gc_eval, read_*, ...), changes of key,
 * schedule or batch geometry, and gc_iknp_*_dev (their column streams advance with every call: the position cannot be
 * baked into a graph; they return GC_E_ARG) are not allowed inside a capture; gc_batch_last_ms is not updated by replays. */
typedef struct gc_graph gc_graph;
int gc_ctx_capture_begin(gc_ctx *);
int gc_ctx_capture_end(gc_ctx *, gc_graph **out);
int gc_graph_launch(gc_graph *);
void gc_graph_free(gc_graph *);

gc_circ *gc_circ_load(gc_ctx *, const gc_gate *gates, uint32_t ngates, uint32_t nwires,
                      uint32_t ninputs, uint32_t noutputs, int *status);
void gc_circ_free(gc_circ *);
const gc_plan *gc_circ_plan(const gc_circ *);
/* execution schedule used by gc_garble / gc_eval on this circuit (see gc_batch_set_schedule):
 * 0 = one launch per dependency level, 1 = fused single launch (default), 2 = fused, single-phase kernel.
 * Bit-identical results. */
int gc_circ_set_schedule(gc_circ *, int schedule);

/* ------------------------------------------------------------------------------------------
 * Host-buffer API — bodies of the reference's per-call entry points, plus a batch dimension
 * (instance-major: instance i's data follows instance i-1's).
 * ------------------------------------------------------------------------------------------ */

/* Replaces (c *Circuit) Garble(rand io.Reader, key []byte) (*Garbled, error)   circuit/garble.go:248
 *  key/keylen  16, 24 or 32 bytes
 *  rnd         per instance the bytes the io.Reader would deliver, in consumption order:
 *              R (16 B; S bit is forced on, garble.go:258) then one L0 per input wire
 *              (garble.go:271-278): stride 16*(1+ninputs); rndlen >= batch*stride or GC_E_RAND
 *  r_out       [batch]                    Garbled.R
 *  wires_out   [batch][nwires] or NULL    Garbled.Wires (both labels, original wire ids)
 *  io_out      [batch][ninputs+noutputs] or NULL: Wires[0:ninputs] then Wires[nwires-noutputs:]
 *              — the only ranges the reference's callers read (garbler.go:87,132,153)
 *  slab_out    [batch][slab_rows]         the dense table slab, gate order (Garbled.Gates[i] =
 *              slab[row_of_gate[i] : row_of_gate[i+1]])                                     */
int gc_garble(gc_circ *, const uint8_t *key, size_t keylen, const uint8_t *rnd, size_t rndlen,
              uint32_t batch, gc_label *r_out, gc_wire *wires_out, gc_wire *io_out, gc_label *slab_out);

/* Replaces (c *Circuit) Eval(key []byte, wires []ot.Label, garbled [][]ot.Label) error   circuit/eval.go:17
 *  wires_inout [batch][nwires] or NULL: inputs pre-filled at [0,ninputs); all wires written in place
 *  inputs      [batch][ninputs]  used when wires_inout is NULL
 *  slab        [batch][slab_rows], slab_rows_given must equal the plan's slab_rows (else GC_E_ROWS)
 *  out_labels  [batch][noutputs] or NULL: wires[nwires-noutputs:]                              */
int gc_eval(gc_circ *, const uint8_t *key, size_t keylen, uint32_t batch, gc_label *wires_inout,
            const gc_label *inputs, const gc_label *slab, size_t slab_rows_given, gc_label *out_labels);

/* The same two calls speaking the wire format of the 2-party driver (SURVEY §8f row 1 on the host-buffer API):
 * gc_garble_wire = Circuit.Garble + the table send loop of circuit.Garbler (circuit/garbler.go:53-82) — wire_out
 * receives, per instance and `stride` bytes apart (multiple of 4, >= gc_tables_wire_bytes()),
 *   BE32(#gates) | per gate BE32(#rows) rows x BE(D0)||BE(D1)
 * i.e. exactly what the SendUint32 / SendLabel loop would put on the connection: the shim hands it to conn in one
 * piece.  gc_eval_wire = the receive loop of circuit.Evaluator (circuit/evaluator.go:40-66) + Circuit.Eval; *bad
 * counts headers that do not match the circuit ("wrong number of gates", evaluator.go:44-47; row counts that Eval
 * rejects, eval.go:54-56,86-89): non-zero returns GC_E_ROWS and nothing is evaluated. */
int gc_garble_wire(gc_circ *, const uint8_t *key, size_t keylen, const uint8_t *rnd, size_t rndlen, uint32_t batch,
                   gc_label *r_out, gc_wire *io_out, uint8_t *wire_out, size_t stride);
int gc_eval_wire(gc_circ *, const uint8_t *key, size_t keylen, uint32_t batch, const gc_label *inputs,
                 const uint8_t *wire_in, size_t stride, gc_label *out_labels, uint32_t *bad);

/* Pinned host memory for the two calls above (additive; the reference pools its scratch on the Go heap,
 * garble.go:195-225 — the shim backs that pool with gc_host_alloc instead).  When slab_out / wires_out / io_out /
 * slab / wires_inout point into pinned memory the calls DMA straight from / into the caller's pages (~57 GB/s instead
 * of 37 / 46 GB/s through the runtime's bounce buffers) and overlap the DMA of one chunk of instances with the
 * layout transpose of the next.  Pageable pointers keep working unchanged.
 *  gc_host_alloc      hipHostMalloc (portable: usable with every device); NULL on failure
 *  gc_host_register   pins an existing range (page-aligned ranges register fastest); undo with gc_host_unregister
 *                     before the memory is freed */
void *gc_host_alloc(size_t bytes);
void gc_host_free(void *);
int gc_host_register(void *p, size_t bytes);
int gc_host_unregister(void *p);
int gc_host_is_pinned(const void *p);

/* Device memory for the device-resident API below (additive; the reference has no device).  A Go host owns no HIP
 * allocator, so the library hands out, fills and reads back the buffers that gc_batch_* (d_rnd, d_bits, d_bits_out,
 * d_mismatch, ...), the *_dev OT calls and gc_comm_allgather take: with these five calls the whole device-resident
 * pipeline is reachable through this header alone (go/circuit/batch_hip.go; tests/cpp/test_device_pipeline.cpp).
 * All of them run on the ctx's device and order themselves on the ctx stream:
 *  gc_dev_alloc     hipMalloc; NULL + *status (GC_E_NOMEM / GC_E_HIP) on failure; contents undefined
 *  gc_dev_free      waits for the ctx stream (kernels may still use the buffer), then frees; NULL is a no-op
 *  gc_dev_upload    host -> device behind everything queued on the ctx stream; returns when `src` may be reused
 *                   (cgo must not leave Go memory referenced after the call)
 *  gc_dev_download  device -> host after everything queued on the ctx stream; returns with the bytes in `dst`
 *  gc_dev_memset    stream-ordered fill, asynchronous
 *  gc_dev_copy      device -> device on the ctx stream, asynchronous (e.g. slot j of an accumulator)
 * d + offset arithmetic on the returned pointers is the caller's (they are plain device addresses). */
void *gc_dev_alloc(gc_ctx *, size_t bytes, int *status);
void gc_dev_free(gc_ctx *, void *d);
int gc_dev_upload(gc_ctx *, void *d_dst, const void *src, size_t bytes);
int gc_dev_download(gc_ctx *, void *dst, const void *d_src, size_t bytes);
int gc_dev_memset(gc_ctx *, void *d, int byte_value, size_t bytes);
int gc_dev_copy(gc_ctx *, void *d_dst, const void *d_src, size_t bytes);

/* Garble ONE instance whose R and input-wire L0 labels are given instead of drawn from a random
 * stream (what Streaming.Garble needs: the inputs of an SSA-step circuit are wires garbled earlier).
 *  r        the stream's R (S bit already set)        inputs   [ninputs] L0 labels
 *  slab_out [slab_rows]                               out_l0   [noutputs] L0 of the output wires */
int gc_garble_labels(gc_circ *, const uint8_t *key, size_t keylen, const gc_label *r, const gc_label *inputs,
                     gc_label *slab_out, gc_label *out_l0);

/* ------------------------------------------------------------------------------------------
 * Streaming garbler (config 5) — circuit/stream_garble.go
 * ------------------------------------------------------------------------------------------ */
typedef struct gc_stream gc_stream;
/* Replaces NewStreaming(cfg, key, inputs, conn)   circuit/stream_garble.go:41-75
 *  rnd = the bytes cfg.GetRandom() would deliver: R (16 B) then one L0 per entry of inputs[] */
gc_stream *gc_stream_create(gc_ctx *, const uint8_t *key, size_t keylen, const uint8_t *rnd, size_t rndlen,
                            const uint32_t *inputs, uint32_t ninputs, int *status);
void gc_stream_free(gc_stream *);
/* Replaces (*Streaming).GetInput(w)   stream_garble.go:117-119 */
int gc_stream_get_wire(gc_stream *, uint32_t w, gc_wire *out);
/* Replaces (*Streaming).Garble(c, in, out)   stream_garble.go:161-192: garbles the circuit (tweak restarts
 * at 0, :174) and appends the serialised gates — op|flags, 16/32-bit wire indexes, table rows, exactly as
 * :391-446 writes them into conn.WriteBuf — to buf.  *written = bytes needed; GC_E_ARG if cap is smaller.
 * in[] / out[] may overlap (output wires that are input wires are resolved through in[] and never set, as in
 * :131-157), and may name the same GLOBAL wire (in-place update): a gate that reads such an input after the gate that
 * set the output sees the new label, as the reference's per-gate stream.wire() look-up does.  Deliberate differences,
 * none reachable from compiled programs: a gate that writes an input-mapped wire is rejected (GC_E_ARG); a gate that
 * reads an output-range wire before any gate of the circuit wrote it is rejected (GC_E_WIRE; the reference would read
 * the global store's current label); a global wire that was never set reads as (L0, L1) = (0, R) — the reference's
 * zero-initialised store gives (0, 0); the engine never stores L1 (always L0 ^ R). */
int gc_stream_garble(gc_stream *, const gc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                     uint32_t nin, const uint32_t *out, uint32_t nout, uint8_t *buf, size_t cap, size_t *written);

/* The same call in two halves (additive): _begin QUEUES one circuit — input labels gathered from the stream's wire store
 * (which lives in HBM), garbling, output labels scattered back, serialisation — and returns without waiting for the GPU;
 * _finish hands out the bytes of the OLDEST circuit in flight.  Calling begin(k + 1 ... k + d) before finish(k) overlaps
 * the host's share of a step (content hash, plan look-up, launches) with the GPU's share of the steps before, and gives
 * the engine STEP-LEVEL PARALLELISM: queued circuits of at most 32 768 gates (8 192 if their levels are wide) that share no global wire through in[] /
 * out[] (no read-after-write, write-after-write, write-after-read) are garbled side by side by ONE launch sequence —
 * one workgroup per circuit — the way independent SSA instructions of a compiled program (compiler/ssa/streamer.go:
 * 412-524 garbles one circuit per instruction) allow; a circuit that depends on a queued one starts the next group, and
 * larger circuits keep a launch sequence of their own.  The bytes still leave in program order, so a driver writes them
 * to the connection exactly as before.  At most 4 096 circuits in flight (a further _begin returns GC_E_ARG);
 * gc_stream_garble is _begin + _finish and needs nothing in flight; gc_stream_get_wire launches what is queued and
 * waits for it.  _finish launches the group of the oldest circuit if it is still open; gc_stream_garble_flush does the
 * same without waiting (a driver that wants the GPU to start before it has queued a whole window).  If cap is too small
 * _finish returns GC_E_ARG with *written = the size needed; that circuit has been garbled (its outputs are set) and its
 * bytes are dropped. */
int gc_stream_garble_begin(gc_stream *, const gc_gate *gates, uint32_t ngates, uint32_t nwires, const uint32_t *in,
                           uint32_t nin, const uint32_t *out, uint32_t nout);
int gc_stream_garble_finish(gc_stream *, uint8_t *buf, size_t cap, size_t *written);
/* The same without the copy (additive): *bytes points into the engine's pinned staging and stays valid until the next
 * gc_stream_garble_finish / _finish_view / gc_stream_free on this stream — for a caller that moves the bytes on itself
 * (the Go shim copies them into conn.WriteBuf, stream_garble.go:177-185: one copy instead of two; on a program of 13 000-gate
 * multipliers the copy out of the staging was most of the host's time per step). */
int gc_stream_garble_finish_view(gc_stream *, const uint8_t **bytes, size_t *len);
/* gc_stream_garble_finish with the copy DEFERRED (additive, round 5): *written is known when the call returns, the bytes are in
 * buf[0 .. *written) once gc_stream_garble_copies_wait (or gc_stream_free) has returned — a few threads of the stream do the
 * copying (GC_STREAM_COPY_THREADS, default 3) while the caller's thread queues the next circuits.  For a host that fills a
 * large write buffer of its own: on the Ed25519-shaped program the stream is 22 bytes per gate, so at 8e8 gates/s the copy
 * out of the staging alone is 17 GB/s — more than the one core that also queues 26 000 circuits can move.  buf must stay
 * valid and untouched until the wait; the three kinds of finish may be mixed (every one hands out the OLDEST circuit). */
int gc_stream_garble_finish_async(gc_stream *, uint8_t *buf, size_t cap, size_t *written);
int gc_stream_garble_copies_wait(gc_stream *);
int gc_stream_garble_flush(gc_stream *);
/* A driver garbles the same few circuits over and over (the streamer keeps one compiled circuit per SSA instruction
 * shape: 23 for Ed25519 sign.mpcl, benchmarks.md:698), and gc_stream_garble_begin has to recognise the gate list by
 * content on every call (a hash and a gate-by-gate comparison: most of the host's share of a small step).
 * gc_stream_intern does that ONCE and returns a handle for the stream's lifetime (the Go shim keeps a
 * map[*Circuit]handle; interned circuits are exempt from cache eviction); gc_stream_garble_begin_h is
 * gc_stream_garble_begin for an interned circuit: in[] / out[] have the circuit's ninputs / noutputs entries. */
int gc_stream_intern(gc_stream *, const gc_gate *gates, uint32_t ngates, uint32_t nwires, uint32_t ninputs,
                     uint32_t noutputs, uint32_t *handle);
int gc_stream_garble_begin_h(gc_stream *, uint32_t handle, const uint32_t *in, const uint32_t *out);
/* gives an interned circuit back to the bounded cache (a host that drops a compiled circuit; the Go shim's
 * (*Streaming).Forget(c) does it — an explicit call: the shim's handle map keeps the *Circuit alive, so no finalizer could):
 * the handle is invalid afterwards (GC_E_ARG) and its number may be handed out again */
int gc_stream_release(gc_stream *, uint32_t handle);
/* launch sequences so far: groups of small circuits, circuits that ran in them, circuits with a sequence of their own
 * (any pointer may be NULL) */
int gc_stream_stats(const gc_stream *, uint64_t *groups, uint64_t *grouped_steps, uint64_t *big_steps);
/* Deep lanes (additive).  A step whose one-workgroup pass is long — hundreds of dependent hash phases: a 128- / 256-bit
 * multiplier, a 256- / 512-bit adder, 0.5 - 2 ms on one CU — runs on one of a few extra HIP streams of the ctx, beside the
 * groups of small steps, and is ordered against them by events only where two steps share a wire; the bytes still leave in
 * program order (circuit/stream_garble.go:161-192 is one serial loop: nothing on the wire or in the wire store may differ
 * from it).  deep_steps: steps that ran that way so far (they are also counted as groups of one by gc_stream_stats);
 * lanes: streams in use (0: none — GC_STREAM_DEEP_LANES=0, or no stream of the runtime runs beside the ctx stream: such
 * steps then keep the launch sequence of the big steps).  GC_STREAM_DEEP_LANES (default 3) and GC_STREAM_DEEP_STEPS
 * (barriers per pass from which a step counts as deep, default 300) are read when a ctx / a stream first needs them.
 * Either pointer may be NULL. */
int gc_stream_deep_stats(const gc_stream *, uint64_t *deep_steps, uint32_t *lanes);
/* Chain fusion (round 5; mpc_amd/csrc/stream_fuse.cpp): a queued step whose dependencies among the steps still queued all sit
 * in ONE launch unit is appended to that unit, and the chain — mul -> add -> add -> ... -> carry — runs as one planned job
 * whose gates are scheduled across the step boundaries (a ten-link chain of 64-bit adders: 63 + 9 dependent hash phases
 * instead of 10 x 63).  The hash tweak still starts over at every circuit (circuit/stream_garble.go:174) and the bytes still
 * leave step by step in program order (:385-449): nothing on the wire or in the wire store differs.  fused_units: launch
 * units of more than one step so far; fused_steps: the steps in them; plans_built: merged plans this stream had to build
 * (they are cached per ctx); unfit: units whose merged plan fits no workgroup (their steps ran one launch after the other).
 * GC_STREAM_NO_FUSE in the environment switches the fusion off.  Any pointer may be NULL. */
int gc_stream_fuse_stats(const gc_stream *, uint64_t *fused_units, uint64_t *fused_steps, uint64_t *plans_built, uint64_t *unfit);
/* Units that WAIT inside a launch — an experiment, off unless GC_STREAM_DEPS=1 is in the environment when the stream is
 * created (EXPERIMENTS.md, round 5: on the programs measured it costs more chain fusion than it saves launches).  A queued step
 * that conflicts with steps of an open group — and cannot be fused into one unit with them: it conflicts with several units,
 * or is too large for a chain — then still joins that group: its workgroup waits, on the device, for the done-flags of exactly
 * the units it conflicts with (mpc_amd/csrc/kernels.h: d_sync) instead of the whole step waiting for the whole group in a
 * later launch.  Program order per wire is what the reference's serial loop gives (stream_garble.go:131-157): a reader runs
 * behind the writer it names, a writer behind the earlier readers and writers of its wire.  waiting_units: such units so far
 * (0 when the experiment is off). */
int gc_stream_wait_stats(const gc_stream *, uint64_t *waiting_units);

/* Streaming evaluator: the store of circuit.StreamEval (stream_evaluator.go:29-96) and the
 * per-gate loop of StreamEvaluator for ONE OpCircuit block (stream_evaluator.go:270-432).  The host driver keeps
 * reading the framing (OpCircuit header: step, numGates, numTmpWires, numWires) and hands the gate bytes over;
 * the gates are parsed back into a circuit, levelised and evaluated with the same kernels as Circuit.Eval.  The wire
 * store lives in HBM: gc_stream_eval_circuit returns once the block is parsed and its kernels are enqueued (the parsing of
 * the next block overlaps them) — or, for blocks of at most 32 768 gates, QUEUED: consecutive small blocks that share no
 * global wire are evaluated side by side by one launch sequence, as on the garbler's side; gc_stream_eval_get_wire
 * launches what is queued and waits for it.  The per-stream circuit cache is bounded (GC_STREAM_CACHE_GATES gates,
 * default 8 Mi; least recently used circuits and their byte skeletons go first): the blocks are the peer's data. */
typedef struct gc_stream_eval gc_stream_eval;
gc_stream_eval *gc_stream_eval_create(gc_ctx *, const uint8_t *key, size_t keylen, int *status);
void gc_stream_eval_free(gc_stream_eval *);
int gc_stream_eval_set_wire(gc_stream_eval *, uint32_t w, const gc_label *l); /* input labels (OT results etc.) */
int gc_stream_eval_get_wire(gc_stream_eval *, uint32_t w, gc_label *l);       /* OpReturn / OpResult reads */
/* *consumed = bytes of buf used by the ngates gates; GC_E_GATE "invalid operation", GC_E_ROWS truncated stream (also:
 * more gates announced than len / 5 bytes can hold), GC_E_ARG a tmp wire read before this block wrote it (tmp wires are
 * private to their OpCircuit block), a tmp id >= ntmp or a global wire id >= nwires, and header sizes no compiler
 * produces: ntmp > 64 ngates + 2^20, nwires > GC_STREAM_MAX_WIRES (default 2^28) — both size arrays.  The block is the peer's data:
 * nothing is sized by it before these checks, and allocation failures come back as GC_E_NOMEM. */
int gc_stream_eval_circuit(gc_stream_eval *, uint32_t ngates, uint32_t ntmp, uint32_t nwires, const uint8_t *buf,
                           size_t len, size_t *consumed);
/* The framing loop of StreamEvaluator around that (stream_evaluator.go:226-270) for as many WHOLE OpCircuit blocks as buf holds
 * (additive).  buf starts at an operation word: 4 bytes big-endian, OpCircuit = 1, then step, numGates, numTmpWires, numWires
 * (compiler/ssa/streamer.go:679-693), then the gates.  Block after block is evaluated as gc_stream_eval_circuit would; the call
 * stops in front of the first operation that is not OpCircuit (OpReturn: the caller reads the result wires) and in front of
 * a block that ends beyond len — then *more = 1: bring more bytes and call again from *consumed, or, if none can come, the
 * stream is truncated (what gc_stream_eval_circuit reports as GC_E_ROWS).  *consumed = bytes used, *nblocks = blocks
 * evaluated (both also when an error is returned: the blocks before the bad one are done); nblocks / more may be NULL.
 * A Go host hands over conn.ReadBuf[ReadStart:ReadEnd] (p2p/protocol.go:29-35, 1 MiB) instead of collecting one block gate
 * by gate; buf is only read during the call.  (A thread comparing the blocks ahead with their byte skeletons while this one
 * queues the ones before was built and measured: +7 % with the whole stream in one call, -25 % in 1 MiB pieces — the rows
 * then reach the queueing thread from another core's cache — and removed.) */
int gc_stream_eval_blocks(gc_stream_eval *, const uint8_t *buf, size_t len, size_t *consumed, uint32_t *nblocks, int *more);
/* Blocks handed to gc_stream_eval_circuit so far: parsed[0] gate by gate, parsed[1] recognised as a block seen before up
 * to its table rows and the global wires it is bound to (same op / flag bytes and tmp ids at the same offsets, global ids
 * repeating in the same pattern): those are not decoded again.  Either pointer may be NULL.  (GC_STREAM_NO_SKELETON in the
 * environment at gc_stream_eval_create time switches the recognition off: every block is decoded.) */
int gc_stream_eval_stats(const gc_stream_eval *, uint64_t *parsed, uint64_t *matched);
/* the evaluator's counterpart of gc_stream_deep_stats */
int gc_stream_eval_deep_stats(const gc_stream_eval *, uint64_t *deep_blocks, uint32_t *lanes);
/* Device-side match (round 5; mpc_amd/csrc/stream_eval_dev.cpp): gc_stream_eval_blocks sends a read buffer of 64 KiB or more
 * to the GPU as it is — one DMA, straight from the caller's memory when that is pinned (gc_host_alloc / gc_host_register), else
 * through pinned staging — where one workgroup recognises the blocks by their byte skeletons and reads their global wire ids;
 * the table rows are gathered from the device copy by the launch that needs them.  The host reads 20 bytes of header and the
 * ids of a block instead of every byte twice.  blocks: blocks recognised that way so far; fallbacks: blocks among them whose
 * ids repeated in another pattern than the skeleton's, parsed by the host after all.  Blocks the device does not know (met
 * for the first time, cut off by the end of the buffer) take the host path as before.  GC_STREAM_NO_DEVICE_MATCH switches it
 * off.  Either pointer may be NULL. */
int gc_stream_eval_dev_stats(const gc_stream_eval *, uint64_t *blocks, uint64_t *fallbacks);
/* the evaluator's counterpart of gc_stream_fuse_stats (its blocks chain exactly as the garbler's steps do) */
int gc_stream_eval_fuse_stats(const gc_stream_eval *, uint64_t *fused_units, uint64_t *fused_blocks, uint64_t *plans_built,
                              uint64_t *unfit);
/* ... and of gc_stream_wait_stats */
int gc_stream_eval_wait_stats(const gc_stream_eval *, uint64_t *waiting_units);

/* ------------------------------------------------------------------------------------------
 * Device-resident batch API — what a Go host pipelining many instances (GarbleBatch/EvalBatch,
 * additive to the reference API) calls; also what bench.py times.  All d_* arguments are
 * DEVICE pointers; calls enqueue on the ctx stream and return without waiting.
 * Device layouts are wire-major / instance-minor so that a wavefront reads 64 consecutive
 * instances of one wire as one 1 KiB coalesced access:
 *   wire labels  uint4 [nslots][bstride]     tables  uint4 [slab_rows][bstride]
 * where bstride = batch rounded up to 64.
 * ------------------------------------------------------------------------------------------ */
typedef struct gc_batch gc_batch;

gc_batch *gc_batch_create(gc_circ *, uint32_t batch, int *status);
void gc_batch_free(gc_batch *);
uint32_t gc_batch_stride(const gc_batch *);
/* geometry the fused schedules chose for this batch: instances per workgroup tile (1 for schedule 0), and
 * whether the live wire labels fit in LDS (otherwise the fused kernel keeps them in HBM) */
uint32_t gc_batch_tile_instances(const gc_batch *);
int gc_batch_wires_in_lds(const gc_batch *);
/* schedule: 0 = one launch per dependency level (the reference's AssignLevels order), labels laid
 *               out [wire][instance];
 *           1 = fused (default): ONE launch per pass, every workgroup owns a tile of instances and
 *               walks all levels with workgroup barriers; labels laid out [tile][wire][instance].  When the
 *               live labels fit in LDS the tile's two halves run one stage apart (one half's XOR chain
 *               hides behind the other half's hashing);
 *           2 = as 1 (same layout) but with the single-phase kernel: every instance of the tile in lock
 *               step (what tiles of one instance use anyway; kept selectable for comparison and tests).
 * Results are bit-identical.  Changing the schedule re-allocates the batch's device arrays. */
int gc_batch_set_schedule(gc_batch *, int schedule);
/* fused schedule only: also write EVERY wire label to the global wire array (needed by
 * gc_batch_read_wires / gc_batch_read_labels; gc_garble / gc_eval switch it on when the caller asks
 * for Garbled.Wires / the full wires slice).  Off by default: only input and output wires are kept. */
int gc_batch_set_store_all(gc_batch *, int on);
/* use a captured hipGraph for the per-level launches (default on) */
int gc_batch_set_graph(gc_batch *, int on);

/* garble all instances: d_rnd = [batch][1+ninputs][16] bytes (same stream as gc_garble) */
int gc_batch_garble(gc_batch *, const uint8_t *key, size_t keylen, const void *d_rnd);
/* evaluator inputs: active label of input wire w of instance i = L0 ^ bit*R, picked on the
 * device from the garbler's state (stands in for "send own labels + OT", garbler.go:85-132);
 * d_bits = u8 [batch][ninputs] */
int gc_batch_select_inputs(gc_batch *evaluator, const gc_batch *garbler, const void *d_bits);
/* evaluator inputs from explicit labels, d_labels = gc_label [batch][ninputs] */
int gc_batch_set_inputs(gc_batch *evaluator, const void *d_labels);
/* evaluate with the tables of `tables` (device layout [slab_rows][bstride]); may be the
 * garbler's batch itself */
int gc_batch_eval(gc_batch *evaluator, const uint8_t *key, size_t keylen, const gc_batch *tables);
/* BitFromLabel (circuit/helpers.go:18-28) for every output wire: d_bits_out = u8 [batch][noutputs];
 * *d_mismatch (u32, device) counts labels that match neither L0 nor L1 */
int gc_batch_decode(const gc_batch *garbler, const gc_batch *evaluator, void *d_bits_out, void *d_mismatch);

/* read-backs (synchronous; host pointers) */
int gc_batch_read_r(gc_batch *, gc_label *r_out);                     /* [batch] */
int gc_batch_read_slab(gc_batch *, gc_label *slab_out);               /* [batch][slab_rows] (reference order) */
int gc_batch_read_wires(gc_batch *, gc_wire *wires_out);              /* garbler: [batch][nwires]; needs store_all */
int gc_batch_read_labels(gc_batch *, gc_label *labels_out);           /* evaluator: [batch][nwires]; needs store_all */
int gc_batch_read_outputs(gc_batch *, gc_label *out);                 /* [batch][noutputs] active/L0 labels */
int gc_batch_write_slab(gc_batch *, const gc_label *slab);            /* host [batch][slab_rows] -> device layout */
/* raw device pointers (for RCCL gathers / the caller's own kernels) */
void *gc_batch_dev_wires(gc_batch *);
void *gc_batch_dev_slab(gc_batch *);
void *gc_batch_dev_r(gc_batch *);
/* output-wire labels gathered into a dense device buffer gc_label [noutputs][bstride] */
int gc_batch_gather_outputs(gc_batch *, void *d_out);
/* Device-side hand-over of input labels through an OT (garbler.go:102-132): the garbler's {L0, L1} pairs of input
 * wires [first, first+count) as gc_wire [batch][count] (what COT.Send consumes), and the evaluator's delivered
 * labels gc_label [batch][count] back into its wire array.  Asynchronous on the ctx stream. */
int gc_batch_gather_input_wires(gc_batch *garbler, uint32_t first, uint32_t count, void *d_wires_out);
int gc_batch_set_input_range(gc_batch *evaluator, uint32_t first, uint32_t count, const void *d_labels);

/* Table egress / ingest in the wire format of the 2-party driver, on the device:
 * replaces the per-gate SendUint32(len) + SendLabel loop of circuit.Garbler (circuit/garbler.go:69-82) and the
 * receive loop of circuit.Evaluator (circuit/evaluator.go:40-66).  Per instance:
 *   BE32(#gates) | per gate: BE32(#rows) rows x BE(D0)||BE(D1)        = gc_tables_wire_bytes() bytes
 * so the host hands the buffer to conn.Write / fills it from the conn in one piece.
 * d_out / d_in: device byte buffers, `stride` bytes between instances (multiple of 4, >= the size).
 * ingest: *d_bad (u32, device) counts headers that do not match the circuit ("wrong number of gates",
 * evaluator.go:44-47; row counts that Eval would reject, eval.go:54-56,86-89). */
size_t gc_tables_wire_bytes(const gc_circ *);
int gc_batch_egress_tables(gc_batch *, void *d_out, size_t stride);
int gc_batch_ingest_tables(gc_batch *, const void *d_in, size_t stride, void *d_bad);
/* sha2pc's encoding of the same tables (sha2pc/encoding.go:363-411 encodeGarbledTables / decodeGarbledTables): the
 * rows of all gates back to back in gate order, BE(D0)||BE(D1) each, no headers: 16 * slab_rows bytes per instance
 * (= garbledTableByteLen, sha2pc/params.go); stride = bytes between instances, a multiple of 16. */
int gc_batch_egress_tables_dense(gc_batch *, void *d_out, size_t stride);
int gc_batch_ingest_tables_dense(gc_batch *, const void *d_in, size_t stride);

/* timing of the most recent garble / eval on this batch, measured with HIP events recorded on
 * the ctx stream around the gate kernels (the one fused launch, or the level launches; the few-microsecond
 * label initialisation kernel of a garble is outside) (ms); negative if none */
float gc_batch_last_ms(gc_batch *);
/* gate-kernel launches issued by the most recent garble / eval (1 for the fused schedules) */
uint32_t gc_batch_last_launches(gc_batch *);
/* developer aid: s_memtime breakdown of the fused kernels.  enable != 0 switches the instrumented
 * build of the kernel on for subsequent passes; out16 (may be NULL) receives, averaged over
 * workgroups, 8 cycle counters of wave 0 {prologue, hash phase, barrier after hash, staging commit,
 * XOR run, chunk-end barrier, -, -} followed by the same 8 for wave 3, from the most recent
 * instrumented pass. */
int gc_batch_debug_profile(gc_batch *, int enable, uint64_t *out8);

/* ------------------------------------------------------------------------------------------
 * IKNP OT extension + MITCCRH (ot/iknp.go, ot/mitccrh.go, ot/cot.go)
 * ------------------------------------------------------------------------------------------ */
typedef struct gc_iknp gc_iknp;

/* Receiver state after the base OTs: replaces the tail of NewIKNPReceiver (iknp.go:347-356):
 * base[128] are the (L0,L1) pairs sent through the base OT; column PRGs g0/g1 are keyed by them */
gc_iknp *gc_iknp_receiver_create(gc_ctx *, const gc_wire *base, int *status);
/* Sender state: replaces the tail of NewIKNPSender (iknp.go:104-122): delta and the 128 labels
 * k0[i] received from the base OT with choice bits delta.Bit(i) */
gc_iknp *gc_iknp_sender_create(gc_ctx *, const gc_label *delta, const gc_label *k0, int *status);
void gc_iknp_free(gc_iknp *);

/* bytes of u-matrix data exchanged for n OTs (sum of the chunk lengths of iknp.go:482-505) */
size_t gc_iknp_u_bytes(size_t n);
/* Replaces the body of (*IKNPReceiver).receive(b, result)   ot/iknp.go:468-511
 *  choice  bool per OT (n bytes, non-zero = true)
 *  u_out   the chunks the Go loop would SendData, concatenated (gc_iknp_u_bytes(n) bytes); the
 *          shim frames them as ≤8 KiB messages exactly like iknp.go:499
 *  labels_out [n]                                                                         */
int gc_iknp_receive(gc_iknp *, const uint8_t *choice, size_t n, uint8_t *u_out, gc_label *labels_out);
/* Replaces the body of (*IKNPSender).send(n)   ot/iknp.go:197-226 (u_in = the received chunks, concatenated) */
int gc_iknp_send(gc_iknp *, const uint8_t *u_in, size_t u_len, size_t n, gc_label *labels_out);
/* Device-resident forms of the two calls above (pipelines that keep the u-matrix and the labels in HBM, e.g.
 * labels that feed gc_batch_set_inputs / the COT kernels directly): all pointers are DEVICE pointers,
 * d_choice_packed = the choice bits packed LSB-first, 64 bytes per 512-OT chunk (iknp.go:472-477; zero-padded to
 * a whole chunk); asynchronous on the ctx stream, no allocation after the first call of a given size.
 * gc_iknp_last_ms: HIP-event time of the kernels of the most recent *_dev call. */
int gc_iknp_receive_dev(gc_iknp *, const void *d_choice_packed, size_t n, void *d_u_out, void *d_labels_out);
int gc_iknp_send_dev(gc_iknp *, const void *d_u_in, size_t n, void *d_labels_out);
float gc_iknp_last_ms(gc_iknp *);

/* bit-COT: bodies of (*IKNPReceiver).ReceiveBits (ot/iknp.go:554-620) and
 * (*IKNPSender).SendBits (ot/iknp.go:259-310); choices / result are packed little-endian u64 bit vectors.
 * The reference only folds WHOLE 64-bit choice words into u (iknp.go:583-597); reproduced as is. */
int gc_iknp_receive_bits(gc_iknp *, const uint64_t *choices, size_t n, uint8_t *u_out, uint64_t *result);
int gc_iknp_send_bits(gc_iknp *, const uint8_t *u_in, size_t u_len, size_t n, uint64_t *result);

/* the same two bodies with the choice words, the u-matrix and the result words in HBM (device pointers, asynchronous on
 * the ctx stream; the labels stay in the handle's device workspace) */
int gc_iknp_receive_bits_dev(gc_iknp *, const void *d_choices, size_t n, void *d_u_out, void *d_result);
int gc_iknp_send_bits_dev(gc_iknp *, const void *d_u_in, size_t n, void *d_result);

/* KOS consistency check of the malicious variant: the chi-PRG + GF(2^128) inner products of
 * (*IKNPReceiver).Receive (ot/iknp.go:405-465) and (*IKNPSender).Send (ot/iknp.go:138-194; gf128.go:14-27,
 * mul128_generic.go).  chi_i = label i of the AES-128-CTR stream keyed by seed2: 0..n-1 for `result`,
 * n..n+255 for the 256-label random choice vector.
 *  receiver: x = XOR_{b_i} chi_i, (t0,t1) = XOR_i chi_i * result_i (256-bit, no reduction)
 *  sender:   *ok = ((q0,q1) ^ x*Delta == (t0,t1)) with (q0,q1) the same sum over its own labels;
 *            ok == 0 is the reference's "OT extension check failed" */
int gc_kos_receiver_tags(gc_ctx *, const gc_label *seed2, const gc_label *result, const uint8_t *b, size_t n,
                         const gc_label *choice_vec, const uint8_t *bcv /*256*/, gc_label *x, gc_label *t0,
                         gc_label *t1);
int gc_kos_sender_check(gc_ctx *, const gc_label *seed2, const gc_label *result, size_t n,
                        const gc_label *choice_vec /*256*/, const gc_label *delta, const gc_label *x,
                        const gc_label *t0, const gc_label *t1, int *ok);
/* the same two checks over labels (and the receiver's choice bytes) that are already in HBM — the outputs of
 * gc_iknp_receive_dev / gc_iknp_send_dev; d_result [n] gc_label, d_b [n] bytes are DEVICE pointers (additive) */
int gc_kos_receiver_tags_dev(gc_ctx *, const gc_label *seed2, const void *d_result, const void *d_b, size_t n,
                             const gc_label *choice_vec, const uint8_t *bcv /*256*/, gc_label *x, gc_label *t0,
                             gc_label *t1);
int gc_kos_sender_check_dev(gc_ctx *, const gc_label *seed2, const void *d_result, size_t n,
                            const gc_label *choice_vec /*256*/, const gc_label *delta, const gc_label *x,
                            const gc_label *t0, const gc_label *t1, int *ok);

/* Replaces (*MITCCRH).Hash over a whole COT/ROT run (mitccrh.go:93-128 as driven by cot.go:160-171,
 * 203-211): OT j (key index gid0+j, key = BE(Label{D0:gid,D1:0} ^ seed)) hashes its h consecutive
 * blocks in place: blk ^= AES_key(blk).  blks = [n][h] */
int gc_mitccrh_hash(gc_ctx *, const gc_label *seed, uint64_t gid0, gc_label *blks, size_t n, uint32_t h);
/* Replaces the pad loop of COT.Send (cot.go:155-182): out[2n] = the labels sent on the wire */
int gc_cot_send_pads(gc_ctx *, const gc_label *seed, const gc_label *delta, const gc_label *data,
                     const gc_wire *wires, size_t n, gc_label *out);
/* Replaces the unpad loop of COT.Receive (cot.go:200-232): result[n] in = IKNP output, out = chosen labels */
/* device-resident forms (device pointers, asynchronous on the ctx stream): d_data = the sender's IKNP labels
 * (gc_iknp_send_dev), d_wires = gc_wire [n], d_out = gc_label [2n]; d_flags = u8 [n] choice bits, d_sent = the 2n
 * labels received, d_result = the receiver's IKNP labels in, the chosen wire labels out */
int gc_cot_send_pads_dev(gc_ctx *, const gc_label *seed, const gc_label *delta, const void *d_data,
                         const void *d_wires, size_t n, void *d_out);
int gc_cot_receive_unpad_dev(gc_ctx *, const gc_label *seed, const void *d_flags, const void *d_sent,
                             void *d_result, size_t n);
int gc_cot_receive_unpad(gc_ctx *, const gc_label *seed, const uint8_t *flags, const gc_label *sent,
                         gc_label *result, size_t n);

/* ROT (ot/rot.go:132-202).  Replaces the pad loop of ROT.Send (rot.go:156-172): data = the sender's IKNP
 * labels (IKNPSender.Send), wires_out[j] = {H_j(data_j), H_j(data_j ^ delta)} — the sender's wire labels are OVERWRITTEN
 * with the two hashed pads (rot.go:168-171), nothing but the seed travels — and the loop of ROT.Receive (rot.go:194-199):
 * result[j] = H_j(result[j]) in place (result in = IKNPReceiver.Receive's labels).  H_j(x) = x ^ AES_{key_j}(x), key_j =
 * BE(Label{D0: j, D1: 0} ^ seed), a fresh MITCCRH per call (rot.go:143,191).  The receiver's result[j] equals
 * wires_out[j].L{flag_j} of the sender.  _dev: device pointers, asynchronous on the ctx stream. */
int gc_rot_send(gc_ctx *, const gc_label *seed, const gc_label *delta, const gc_label *data, size_t n, gc_wire *wires_out);
int gc_rot_receive(gc_ctx *, const gc_label *seed, gc_label *result, size_t n);
int gc_rot_send_dev(gc_ctx *, const gc_label *seed, const gc_label *delta, const void *d_data, size_t n, void *d_wires_out);
int gc_rot_receive_dev(gc_ctx *, const gc_label *seed, void *d_result, size_t n);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU (BASELINE config 4: 65 536 instances, 8 192 per GPU): instances are independent
 * (fresh R and labels per Garble call, circuit/garble.go:253-278), so each device garbles / evaluates a contiguous
 * instance range through its own gc_ctx and the ONLY exchange is the terminal all-gather of the decoded output
 * bits (gc_batch_decode) or output labels (gc_batch_gather_outputs) — ncclAllGather of RCCL over xGMI, enqueued on
 * the ctx stream behind the kernels that produce the data.  The reference has no counterpart (single device); a Go
 * host (apps/garbled, circuit/garbler.go:53) calls these next to GarbleBatch / EvalBatch.  librccl is opened on first
 * use; without it every call returns GC_E_HIP (gc_last_error() says why) and the single-GPU path is unaffected.
 * ------------------------------------------------------------------------------------------ */
typedef struct gc_comm gc_comm;
#define GC_COMM_ID_BYTES 128
int gc_comm_available(void); /* 1 if librccl could be opened and has every entry point used here */
int gc_comm_version(void);   /* ncclGetVersion code (e.g. 22707), 0 if unavailable */
/* One process (or OS thread) per GPU: rank 0 draws the id (ncclGetUniqueId), the host hands the 128 bytes to the other
 * ranks over its own control channel, every rank joins with its ctx (ncclCommInitRank; collective, blocks until all
 * nranks have called). */
int gc_comm_get_unique_id(uint8_t *id, size_t len /* >= GC_COMM_ID_BYTES */);
gc_comm *gc_comm_init_rank(gc_ctx *, const uint8_t *id, size_t idlen, int nranks, int rank, int *status);
/* One process driving n devices (one gc_ctx each, distinct devices): ncclCommInitAll; out[n] */
int gc_comm_init_all(gc_ctx *const *ctxs, int n, gc_comm **out);
void gc_comm_destroy(gc_comm *);
int gc_comm_rank(const gc_comm *);
int gc_comm_nranks(const gc_comm *);
/* d_recv[r * bytes .. (r+1) * bytes) = rank r's d_send[0 .. bytes): device pointers, `bytes` equal on all ranks
 * (the host pads the last shard); asynchronous on the ctx stream */
int gc_comm_allgather(gc_comm *, const void *d_send, void *d_recv, size_t bytes);
/* the same for the n communicators of gc_comm_init_all from ONE host thread (ncclGroupStart / ncclGroupEnd) */
int gc_comm_allgather_all(gc_comm *const *comms, int n, const void *const *d_send, void *const *d_recv, size_t bytes);
/* host-side helpers for drivers and the benchmark: *value = max over ranks (synchronous); barrier = every rank's ctx
 * stream has drained */
int gc_comm_allreduce_max(gc_comm *, double *value);
int gc_comm_barrier(gc_comm *);

#ifdef __cplusplus
}
#endif
#endif /* GCENGINE_H */
