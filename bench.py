#!/usr/bin/env python3
"""bench.py — AND-gates/s (garble+eval) of the AES-128 circuit batch on MI355X.

One "step" = one pass of the hot path over one batch of synthetic instances that are already
resident in HBM: Circuit.Garble for `batch` instances, hand-over of the evaluator's input labels,
Circuit.Eval, and BitFromLabel decoding (+ the RCCL all-gather of the decoded outputs when N > 1: gc_comm_allgather of
libgcengine.so = ncclAllGather over xGMI called from the C ABI).  No torch in this process: device buffers come from
gc_dev_alloc / gc_dev_upload of the same C ABI a Go host binds, `python -m torch.distributed.run` is only the process
launcher, and the 128-byte communicator id travels through a file (mpc_amd/dist.py).
Workload at N=1: BASELINE.json configs[1] — aes_128 (36 663 gates, 6 400 AND) x 1 024 instances,
32-byte garbling key (AES-256, as circuit.Garbler uses).  N > 1: BASELINE.json configs[3]'s shape — 8 192 instances per
GPU on every rank (65 536 at N = 8; weak scaling, independent instances, circuit/garble.go:253-278 draws fresh R and labels
per instance), no data-path collective except the gather of the decoded outputs; `--batch` overrides either.
Every wait of the N > 1 path is bounded (communicator id hand-over, ncclCommInitRank, the run itself: mpc_amd/dist.py
Watchdog); on any failure rank 0 prints ONE JSON line with "error" and the process exits non-zero instead of hanging.

Prints ONE JSON line on rank 0 (see the contract in the task description).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per gate per side (SURVEY.md §8d / DESIGN.md): 16-byte labels, L1 never stored
ALG_BYTES = {"xor": 48, "xnor": 48, "and": 80, "inv": 48, "or": 96}
GATHER_EVERY = 8  # steps per all_gather of the decoded outputs (N > 1)
PREWARM_STEPS = 100  # untimed launches of the step ahead of the warm-up steps: the GPU at its sustained clocks (see run())
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
# Secondary bound (SURVEY.md §8d "LDS bandwidth / VALU issue on AND-dense levels"): the fixed-key hash is a
# T-table AES out of LDS.  tools/aes_ubench measures the production AES core alone at 10.28 cycles per block per
# CU with 16 waves/CU (59.7 G blocks/s on 256 CUs; profiles/r01_aes_ubench.txt).
AES_CORE_PEAK_BLOCKS = 59.7e9
AES_BLOCKS = {"and": (4, 2), "inv": (2, 1), "or": (4, 1)}  # (garble, eval) distinct AES blocks per gate
# bytes READ per gate (garble, eval) of the same layout-independent model (393.4 B per AND for aes_128):
# the north star's "HBM-read roofline"
READ_BYTES = {"xor": (32, 32), "xnor": (32, 32), "and": (32, 64), "inv": (16, 32), "or": (32, 80)}
# LDS array: a conflict-free wave64 ds_read_b32 takes 2 LDS cycles = 128 B/clk/CU, ~75 TB/s with every CU streaming
# (MI355X_MICROARCH.md, LDS section) = 18.75 T table look-ups/s; one AES block is 16 look-ups per round.
LDS_B32_PEAK_LOOKUPS = 75e12 / 4


def alg_bytes_per_instance(info):
    return (info.n_xor * ALG_BYTES["xor"] + info.n_xnor * ALG_BYTES["xnor"] + info.n_and * ALG_BYTES["and"] +
            info.n_inv * ALG_BYTES["inv"] + info.n_or * ALG_BYTES["or"])


def cpu_baseline(circ, key, seconds=12.0):
    """The oracle (CPU restatement of the reference's serial Garble/Eval loops, AES-NI) timed on the
    host cores: one instance per thread-iteration, all cores.  Reported, never the target."""
    import oracle

    # threads = the CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes
    # show 256 hardware threads and a quota of 16 CPUs: 256 threads there are throttled to a third of what 16 achieve)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    host_threads = threads
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            threads = max(1, min(threads, int(q) // int(per)))
    except (OSError, ValueError):
        pass
    probe = 40 * threads
    dt = oracle.bench_garble_eval(circ.Gates, circ.NumWires, circ.num_inputs, circ.num_outputs, key, probe, threads)
    reps = max(probe, int(probe * seconds / max(dt, 1e-3)))
    dt = oracle.bench_garble_eval(circ.Gates, circ.NumWires, circ.num_inputs, circ.num_outputs, key, reps, threads)
    ands = circ.stats()["AND"]
    # one thread, the apples-to-apples counterpart of the reference's serial Go loop (~2 s sample)
    r1 = max(8, int(2.0 * reps / max(dt, 1e-3) / threads))
    dt1 = oracle.bench_garble_eval(circ.Gates, circ.NumWires, circ.num_inputs, circ.num_outputs, key, r1, 1)
    out = {
        "value": reps * ands / dt,
        "unit": "AND-gates/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d instances of %s garble+eval, oracle C loop (AES-NI=%s), %d threads (CPU quota of the container; "
                  "the host shows %d hardware threads), %.1f s" % (
            reps, circ.name or "circuit", oracle.using_aesni(), threads, host_threads, dt),
        "single_thread_value": r1 * ands / dt1,
        "published_reference": "155.1 ns/AND garble-only = 6.45 M AND/s, Go, i5-8257U 1 thread (benchmarks.md:726)",
    }
    ref = os.path.join(ROOT, "oracle", "_ref", "aesni_bench")
    if os.path.exists(ref):
        try:
            import subprocess
            txt = subprocess.run([ref], capture_output=True, text=True, timeout=60).stdout
            for ln in txt.splitlines():
                if ln.startswith("AES-NI+C"):
                    out["reference_aesni_c_ns_per_encrypt"] = float(ln.split()[-2])
        except Exception:
            pass
    return out


def kernel_build_hash(circuit_path=None):
    """What the PMC counters of profiles/latest_pmc.json are tied to: a SHA-256 over (i) the DEVICE sources the garble / eval
    kernels are compiled from and (ii) the fingerprint of the benchmarked circuit's plan (gc_plan_fingerprint: the device
    program the planner emits for it).  A change to the host side of the planner that leaves this circuit's program alone
    keeps the counters valid (round 5 hashed plan.cpp's text, and a late planner commit for OTHER circuits nulled
    `roofline.traffic` in the driver's line); a change to the kernels or to this circuit's schedule drops them."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "mpc_amd", "csrc")
    for name in ("aes_device.h", "kernels.h", "plan.h", "level_gate.h", "fused_flat_kernels.hip", "fused_lds_kernels.hip",
                 "fused_kernels.hip", "gc_kernels.hip"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    from mpc_amd import engine, parse_file
    c = parse_file(circuit_path or os.path.join(ROOT, "tests", "golden", "aes_128.gcf"))
    h.update(b"plan\0" + engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs).fingerprint().encode())
    return h.hexdigest()[:16]


SWEEP_KEEP = ("circuit", "gates", "and", "levels", "gates_materialised", "wires_in_lds", "garble_ms", "eval_ms", "and_gates_per_s",
              "gates_per_s", "hbm_alg_GBs", "hbm_roofline_frac", "hbm_read_roofline_frac", "lds_array_frac", "model_note",
              "outputs_ok")
REF_BENCH_KEY = b"0123456789abcdef"  # benchKey of circuit/garble_bench_test.go:35


def sweep_rows_for_line(batch, key, ctx):
    """SURVEY §8d's synthetic levelised circuits for the default line: W = 1 024 at f in {0, 0.17, 1} and one row each of the
    narrow (W = 64) and the wide (W = 16 384) end at the AES-like f = 0.17 (`python bench.py --sweep` runs the whole grid)"""
    from scripts.sweep_synthetic import run as sweep_run
    rows = sweep_run(batch, 131072, key, ctx=ctx, cases=[(1024, 0.0), (1024, 0.17), (1024, 1.0), (64, 0.17), (16384, 0.17)],
                     chain=0)
    return [{k: r[k] for k in SWEEP_KEEP if k in r} for r in rows]


def per_gpu_share_row(batch, circ, key, ctx, steps=20):
    """config 4's per-GPU share on ONE GPU (aes_128 x 8 192 instances, the production schedule, the step in a hipGraph): the
    same-run N = 1 baseline of a scaling curve whose N > 1 points run 8 192 instances per rank."""
    import numpy as np

    from mpc_amd import engine
    dc = engine.DeviceCircuit(ctx, circ)
    info = dc.info
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    for b in (gb, ev):
        b.set_graph(True)
        b.set_schedule(1)
    d_rnd = ctx.random_u8((batch, circ.num_inputs + 1, 16), 256, seed=91)
    d_bits = ctx.random_u8((batch, circ.num_inputs), 2, seed=92)
    d_out = ctx.zeros((batch, circ.num_outputs))
    d_mis = ctx.zeros(1, np.int32)

    def step():
        gb.garble(key, d_rnd)
        ev.select_inputs(gb, d_bits)
        ev.eval(key, gb)
        gb.decode(ev, d_out, d_mis)

    step()
    ctx.sync()
    try:
        graph = ctx.capture(step)
        launch = graph.launch
    except Exception:  # (capture is an optimisation)
        ctx.sync()
        launch = step
    for _ in range(12):
        launch()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        launch()
    ctx.sync()
    dt = time.perf_counter() - t0
    row = {"workload": "%s x %d instances (BASELINE config 4's share of one GPU), %d-byte key" % (circ.name, batch, len(key)),
           "steps": steps, "ms_per_step": dt / steps * 1e3, "and_gates_per_s": info.n_and * batch * steps / dt,
           "graph": launch is not step, "outputs_ok": int(d_mis.numpy()[0]) == 0}
    gb.close()
    ev.close()
    dc.close()
    for d in (d_rnd, d_bits, d_out, d_mis):
        d.close()
    return row


def level_launch_row(batch, circ, key, ctx):
    """The north star's literal schedule beside the production one (SURVEY §7 hard part 4): schedule 0 —
    one data-parallel launch per dependency level (308 garbling + 308 evaluating launches for aes_128, thread = (gate,
    instance), wires in HBM), the step's launches recorded in a hipGraph — on the headline's workload, timed like the
    headline (wall time between two device syncs) with the passes' HIP-event times beside it."""
    import numpy as np

    from mpc_amd import engine
    dc = engine.DeviceCircuit(ctx, circ)
    info = dc.info
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    for b in (gb, ev):
        b.set_schedule(0)
        b.set_graph(True)
    d_rnd = ctx.random_u8((batch, circ.num_inputs + 1, 16), 256, seed=77)
    d_bits = ctx.random_u8((batch, circ.num_inputs), 2, seed=78)
    d_out = ctx.zeros((batch, circ.num_outputs))
    d_mis = ctx.zeros(1, np.int32)

    def step():
        gb.garble(key, d_rnd)
        ev.select_inputs(gb, d_bits)
        ev.eval(key, gb)
        gb.decode(ev, d_out, d_mis)

    for _ in range(3):
        step()
    ctx.sync()
    steps = 6
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    dt = time.perf_counter() - t0
    g_ms, e_ms = [], []
    for _ in range(4):
        gb.garble(key, d_rnd)
        g_ms.append(gb.last_ms)
        ev.select_inputs(gb, d_bits)
        ev.eval(key, gb)
        e_ms.append(ev.last_ms)
    ctx.sync()
    algb = alg_bytes_per_instance(info)
    row = {"schedule": "level-launch (schedule 0): one launch per dependency level, recorded in a hipGraph",
           "workload": "%s x %d instances, %d-byte key" % (circ.name, batch, len(key)),
           "launches_per_garble": int(gb.last_launches), "levels": int(info.nlevels), "steps": steps,
           "ms_per_step": dt / steps * 1e3, "and_gates_per_s": info.n_and * batch * steps / dt,
           "garble_ms": float(np.mean(g_ms)), "eval_ms": float(np.mean(e_ms)),
           "hbm_alg_GBs_garble": algb * batch / (float(np.mean(g_ms)) * 1e-3) / 1e9,
           "hbm_roofline_frac_garble": algb * batch / (float(np.mean(g_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "outputs_ok": int(d_mis.numpy()[0]) == 0}
    gb.close()
    ev.close()
    dc.close()
    for d in (d_rnd, d_bits, d_out, d_mis):
        d.close()
    return row


def reference_bench_rows(batch, circ, ctx):
    """SURVEY §8d's remaining rows, under the reference's own benchmark key (circuit/garble_bench_test.go:35, 16 bytes =
    AES-128): config 2 again (`key16`) and buildANDChain(10000) (:19-33, :39 — depth 10 000, width 1: the worst case)"""
    from mpc_amd.circuit import and_chain
    from scripts.sweep_synthetic import run as sweep_run
    rows = sweep_run(batch, 131072, REF_BENCH_KEY, ctx=ctx, cases=[], chain=0, circuits=[circ, and_chain(10000)])
    keep = SWEEP_KEEP + ("hash_phases", "live_labels")
    k16, chain = ({k: r[k] for k in keep if k in r} for r in rows)
    t = (k16["garble_ms"] + k16["eval_ms"]) * 1e-3
    k16.update({"key_bytes": 16, "value": k16["and_gates_per_s"], "ms_per_step_device": t * 1e3,
                "note": "device time of garble + eval per pass (HIP events), without the label hand-over and decode kernels "
                        "of the headline's step"})
    chain.update({"key_bytes": 16, "ns_per_and_per_instance_garble": chain["garble_ms"] * 1e6 / 10000,
                  "published_reference": "BenchmarkGarble on this circuit: 155.1 ns per AND, one instance, i5-8257U "
                                         "(benchmarks.md:726)"})
    return {"key16": k16, "and_chain_10000": chain}


METRIC = "AND-gates/sec (garble+eval), AES-128 circuit batch"


def error_line(msg, stage, world):
    """the ONE line of a failed run: same metric, no value, what went wrong and where"""
    return json.dumps({"metric": METRIC, "value": None, "unit": "AND-gates/s", "n_gpus": world, "error": str(msg)[:600],
                       "stage": stage, "higher_is_better": True})


def main():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    stage = ["start"]
    try:
        run(stage)
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 — a failed rank must say so and leave; the launcher ends the others
        sys.stderr.write("bench.py rank %d failed at stage '%s': %r\n" % (rank, stage[0], e))
        if rank == 0:
            print(error_line("%s: %s" % (type(e).__name__, e), stage[0], world), flush=True)
        sys.stdout.flush()
        os._exit(1)  # (not sys.exit: a wedged collective must not be waited for by destructors)


def run(stage):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (default: 1 024 at N = 1 = BASELINE config 2, "
                    "8 192 at N > 1 = config 4's per-GPU share)")
    ap.add_argument("--init-timeout", type=float, default=240.0, help="N > 1: seconds the communicator may take to form")
    ap.add_argument("--run-timeout", type=float, default=1500.0, help="N > 1: seconds the whole run may take")
    ap.add_argument("--circuit", default=os.path.join(ROOT, "tests", "golden", "aes_128.gcf"))
    ap.add_argument("--key-bytes", type=int, default=32, choices=[16, 24, 32])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-iknp", action="store_true", help="skip the IKNP OT-extension side measurement")
    ap.add_argument("--no-host-api", action="store_true", help="skip the PCIe-inclusive gc_garble / gc_eval side measurement")
    ap.add_argument("--no-prewarm", action="store_true", help="no untimed launches ahead of the warm-up steps (the GPU's clocks then ramp inside the timed region)")
    ap.add_argument("--no-stream", action="store_true", help="skip the streaming (config 5 shape) side measurement")
    ap.add_argument("--no-config3", action="store_true", help="skip the sha256xor x 256 + 65 536 OTs pipeline (config 3) side measurement")
    ap.add_argument("--no-synthetic", action="store_true", help="skip the five synthetic levelised rows (SURVEY §8d)")
    ap.add_argument("--no-extra-rows", action="store_true", help="skip config 2 under the reference's 16-byte benchmark key "
                    "and the AND-chain row (circuit/garble_bench_test.go:19-39)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--force-collective", action="store_true", help="run the output gather even with one rank (testing)")
    ap.add_argument("--schedule", type=int, default=1)
    ap.add_argument("--check", action="store_true", help="verify decoded outputs against plaintext evaluation")
    ap.add_argument("--sweep", action="store_true", help="synthetic levelised circuits (SURVEY §8d): W x AND-fraction grid + "
                    "AND chain, AND/s and roofline fractions per circuit, one JSON line")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py --gpus %d must be launched through torch.distributed.run" % args.gpus, file=sys.stderr)
            sys.exit(2)
        args.gpus = world

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes); before the runtime starts
    import importlib

    import numpy as np

    from mpc_amd import dist as gdist, parse_file
    # (GC_BENCH_ENGINE: the CPU suite runs this very rank entry at world 2 on a stand-in engine module, tests/stub_engine.py)
    engine = importlib.import_module(os.environ.get("GC_BENCH_ENGINE", "mpc_amd.engine"))

    if args.batch is None:
        args.batch = 1024 if world == 1 else 8192
    circ = parse_file(args.circuit)
    circ.name = os.path.splitext(os.path.basename(args.circuit))[0]
    key = bytes(range(args.key_bytes))
    # N > 1: nothing may hang the lease — a watchdog thread ends the process with the error line when a stage overruns
    # (the main thread may be inside ncclCommInitRank or a collective, i.e. inside a C call that never returns)
    dog = gdist.Watchdog(lambda what: (print(error_line(what, stage[0], world), flush=True) if rank == 0 else None)) if world > 1 else None
    stage[0] = "context"
    # one process per GPU: this rank's device, its own HIP stream (GC_BENCH_DEVICE: a probe that puts every rank on one
    # device to see how far the N > 1 path gets on a one-GPU box — RCCL refuses such a communicator)
    # (a launcher that gives every rank its own HIP_VISIBLE_DEVICES shows each process ONE device, number 0: the local rank is
    # taken modulo what this process can see)
    ndev = engine.device_count() if hasattr(engine, "device_count") else 0
    ctx = engine.Context(int(os.environ.get("GC_BENCH_DEVICE", local_rank % ndev if ndev > 0 else local_rank)))
    collective = world > 1 or args.force_collective
    comm = None
    if collective:  # gc_comm_init_rank: RCCL, one rank per GPU
        stage[0] = "communicator (id hand-over + ncclCommInitRank)"
        if dog:
            dog.arm(args.init_timeout, "no communicator of %d ranks within %.0f s" % (world, args.init_timeout))
        comm = gdist.open_comm(ctx, rank, world, timeout=args.init_timeout * 0.75, engine=engine)
        if comm.nranks != world:
            raise RuntimeError("communicator has %d ranks, launched %d" % (comm.nranks, world))
        if dog:
            dog.arm(args.run_timeout, "the run did not finish within %.0f s" % args.run_timeout)
    stage[0] = "set-up"
    if args.sweep:  # every rank sweeps its own GPU (independent instances); rank 0 reports the job
        from scripts.sweep_synthetic import run as sweep_run
        t0 = time.perf_counter()
        rows = sweep_run(args.batch, 131072, key, ctx=ctx)
        dt = time.perf_counter() - t0
        if comm is not None:
            comm.barrier()
            comm.close()
        if rank == 0:
            for r in rows:
                for k in ("and_gates_per_s", "nonfree_gates_per_s", "gates_per_s", "hbm_alg_GBs"):
                    r[k + "_job"] = r[k] * world  # weak scaling: the same sweep on every GPU
            print(json.dumps({"metric": "AND-gates/sec (garble+eval), synthetic levelised circuits", "unit": "AND-gates/s",
                              "n_gpus": world, "scaling": "weak", "data": "synthetic", "dtype": "u32 (AES T-table / label XOR)",
                              "config": {"workload": "SURVEY §8d grid W in {64,1024,16384} x f in {0,0.17,0.5,1} + AND chain, "
                                         "131072 gates, batch=%d per GPU, %d-byte key" % (args.batch, args.key_bytes)},
                              "sweep_wall_s": dt, "sweep": rows}), flush=True)
        ctx.close()
        return
    dc = engine.DeviceCircuit(ctx, circ)
    info = dc.info
    batch = args.batch
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    for b in (gb, ev):
        b.set_graph(not args.no_graph)
        b.set_schedule(args.schedule)

    # synthetic inputs, resident in HBM before the timed region (gc_dev_alloc + gc_dev_upload: the C ABI's own
    # device memory, what a Go host uses)
    nout = circ.num_outputs
    d_rnd = ctx.random_u8((batch, circ.num_inputs + 1, 16), 256, seed=1234 + rank)
    d_bits = ctx.random_u8((batch, circ.num_inputs), 2, seed=4321 + rank)
    # Decoded outputs of GATHER_EVERY steps are collected in one accumulator and gathered with ONE all-gather (fewer,
    # larger collectives: 1 MiB per GPU per call at 1 024 instances; a gather per step cost ~3 %).  The gather is
    # enqueued on the engine's stream right behind the decode that fills the last slot (stream order protects the
    # accumulator; no host synchronisation inside the loop).
    K = GATHER_EVERY if collective else 1
    slot_bytes = batch * nout
    d_acc = ctx.zeros((K, batch, nout))
    d_mis = ctx.zeros(1, np.int32)
    d_all = ctx.zeros((world, K, batch, nout)) if collective else None

    def device_step(j=0):
        gb.garble(key, d_rnd)
        ev.select_inputs(gb, d_bits)
        ev.eval(key, gb)
        gb.decode(ev, d_acc + j * slot_bytes, d_mis)

    graphs = None  # the step's kernels recorded once per output slot in a hipGraph (gc_ctx_capture_*): one launch per step

    def launch(j):
        if graphs is not None:
            graphs[j].launch()
        else:
            device_step(j)

    def gather(nfresh):  # the only collective: ncclAllGather of the decoded outputs over RCCL/xGMI, behind the C ABI
        comm.allgather(d_acc, d_all, d_acc.nbytes)

    def fence():
        ctx.sync()
        if comm is not None:
            comm.barrier()  # every rank's engine stream has drained (allreduce + stream sync)

    device_step()  # first call uploads the round keys (not capturable), and warms the allocator
    ctx.sync()
    if not args.no_graph and args.schedule != 0:
        try:
            graphs = [ctx.capture(lambda j=j: device_step(j)) for j in range(K)]
        except Exception as e:  # capture is an optimisation: fall back to direct launches of the same kernels
            print("bench: hipGraph capture unavailable (%s); launching directly" % e, file=sys.stderr)
            graphs = None
            ctx.sync()
    # The GPU's clocks fall back after ~10 ms without work and take a few ms of load to come up again (scripts/probe_fixed_cost.py:
    # 20 graph launches behind a 20 ms pause take 1.4 ms longer than behind a 2 ms pause; the set-up above — circuit upload,
    # graph capture — leaves the GPU idle for far longer, and W = 5 warm-up steps are 5 ms).  Untimed launches of the same step
    # bring it to its sustained state before the warm-up steps; nothing is waited for in between, so the warm-up starts behind them.
    if not args.no_prewarm:
        for _ in range(PREWARM_STEPS):
            launch(0)
    stage[0] = "timed steps"
    loop = gdist.StepLoop(K, launch, gather if collective else None)
    elapsed = gdist.run_timed(loop, fence, args.steps, args.warmup,
                              allreduce_max=comm.allreduce_max if comm is not None else None)
    assert loop.steps_gathered == loop.steps_done == args.steps + args.warmup
    stage[0] = "after the timed steps"
    gather_us = None
    if collective:  # the gather alone: a few calls between device syncs (config 4's "RCCL gather over xGMI")
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            gather(K)
        ctx.sync()
        gather_us = (time.perf_counter() - t0) / 5 * 1e6

    # per-pass device times (events on the engine stream) from a few extra, untimed-by-wall steps
    g_ms, e_ms = [], []
    for _ in range(min(20, max(1, args.steps))):
        gb.garble(key, d_rnd)
        g_ms.append(gb.last_ms)
        ev.select_inputs(gb, d_bits)
        ev.eval(key, gb)
        e_ms.append(ev.last_ms)
    ctx.sync()
    mismatches = int(d_mis.numpy()[0])

    ok = mismatches == 0
    acc = d_acc.numpy()
    if collective:  # the gathered array holds this rank's outputs at its offset
        ok = ok and bool((d_all.numpy()[rank] == acc).all())
    if args.check:
        bits = d_bits.numpy()
        for i in range(0, batch, max(1, batch // 16)):  # against plaintext evaluation (circuit/computer.go)
            plain = circ.compute_bits(bits[i])
            ok = ok and bool((plain[circ.NumWires - nout:] == acc[0][i]).all())

    n_and = info.n_and
    total_and = n_and * batch * world * args.steps
    value = total_and / elapsed
    algb = alg_bytes_per_instance(info)
    g_avg = float(np.mean(g_ms))
    e_avg = float(np.mean(e_ms))
    launches = gb.last_launches
    # dominant kernel: the garble pass (schedule 1: ONE k_garble_flat launch; schedule 0: one
    # k_garble_level launch per level).  Algorithmic bytes per launch = garble bytes per instance x batch
    # / launches; launch duration from HIP events recorded on the engine's stream (gc_batch_last_ms).
    achieved = algb * batch / (g_avg * 1e-3) / 1e9
    kernel_name = {0: "k_garble_level", 1: "k_garble_flat", 2: "k_garble_lds"}[args.schedule]
    sched_name = {0: "level-launch", 1: "fused-flat", 2: "fused-levels"}[args.schedule]
    blocks_g = sum(getattr(info, "n_" + k) * v[0] for k, v in AES_BLOCKS.items())
    blocks_e = sum(getattr(info, "n_" + k) * v[1] for k, v in AES_BLOCKS.items())
    # HBM traffic of the garble launch: PMC counters cannot be read from inside the run; the figure is the one
    # scripts/profile.sh measured with rocprofv3 --pmc for this batch / schedule / key size AND this build of the
    # kernels (profiles/latest_pmc.json carries the hash of the kernel sources) — null otherwise.
    traffic, traffic_src = None, None
    build_hash = kernel_build_hash(args.circuit)
    pmc = os.path.join(ROOT, "profiles", "latest_pmc.json")
    if os.path.exists(pmc):
        try:
            with open(pmc) as f:
                pj = json.load(f)
            if (pj.get("batch") == batch and pj.get("schedule") == args.schedule and pj.get("key_bytes") == args.key_bytes
                    and pj.get("kernel_build_hash") == build_hash):
                traffic = pj.get("garble_hbm_bytes_per_launch")
                traffic_src = "profiles/latest_pmc.json (rocprofv3 --pmc, %s)" % pj.get("source", "scripts/profile.sh")
            elif pj.get("kernel_build_hash") != build_hash:
                traffic_src = "null: profiles/latest_pmc.json was measured on kernel build %s, this is %s" % (
                    pj.get("kernel_build_hash"), build_hash)
        except Exception:
            traffic = None
    read_b = sum(getattr(info, "n_" + k) * (v[0] + v[1]) for k, v in READ_BYTES.items())  # per instance, both passes
    rounds = {16: 10, 24: 12, 32: 14}[args.key_bytes]
    lookups_per_block = 16 * rounds
    lds_g = blocks_g * batch * lookups_per_block / (g_avg * 1e-3) / LDS_B32_PEAK_LOOKUPS
    lds_e = blocks_e * batch * lookups_per_block / (e_avg * 1e-3) / LDS_B32_PEAK_LOOKUPS
    frac_read = value / world * (read_b / max(n_and, 1)) / 1e9 / HBM_PEAK_GBS
    res = {
        "metric": METRIC,
        "value": value,
        "unit": "AND-gates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 (AES T-table / 128-bit label XOR, integer)",
        "data": "synthetic (uniform random label streams and input bits, resident in HBM)",
        "config": {
            "workload": "%s.circ %d gates / %d AND, batch=%d instances per GPU, %d-byte garbling key" % (
                circ.name, info.ngates, n_and, batch, args.key_bytes),
            "instances_per_gpu": batch,
            "levels": int(info.nlevels),
            "launches_per_garble": int(launches),
            "schedule": sched_name,
            "hash_phases": int(info.n_hash_phases),
            "lds_live_labels": int(info.n_flat_slots if args.schedule == 1 else info.n_lds_slots),
            "graph": graphs is not None or (args.schedule == 0 and not args.no_graph),
            "outputs_ok": ok,
            "device_memory": "gc_dev_alloc / gc_dev_upload (C ABI); no torch in the process",
            "gathers": loop.gathers,
        },
        # what the communicator itself says (gc_comm_nranks / gc_comm_version), not what the launcher promised
        "n_ranks_seen": comm.nranks if comm is not None else 1,
        "rccl_version": engine.comm_version() if comm is not None else None,
        "garble_ms": g_avg,
        "eval_ms": e_avg,
        "and_gates_per_s_garble_only": n_and * batch / (g_avg * 1e-3),
        "and_gates_per_s_eval_only": n_and * batch / (e_avg * 1e-3),
        "hbm_alg_GBs_garble_plus_eval": 2 * algb * batch / ((g_avg + e_avg) * 1e-3) / 1e9,
        "roofline": {
            # what binds the dominant kernel: the T-table AES out of LDS (LDS array + VALU issue), not HBM — the wires
            # never leave the CU.  achieved / peak / frac keep the contract's definition (SURVEY §8d algorithmic bytes of
            # the garble launch over its HIP-event duration against the 8 TB/s HBM peak): a NOMINAL figure, restated as
            # frac_nominal_model; frac_read is the north star's "HBM-read roofline" of the whole step; frac_bound prices
            # the kernel against the resource that binds it.
            "bound": "lds/valu-issue" if args.schedule != 0 else "hbm",
            "kernel": kernel_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_nominal_model": achieved / HBM_PEAK_GBS,
            "frac_read": frac_read,
            "frac_bound": lds_g,
            "frac_bound_definition": "T-table look-ups/s of the garble launch over the LDS array's ds_read_b32 rate "
                                     "(75 TB/s / 4 B, MI355X_MICROARCH.md)",
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel_build_hash": build_hash,
            "alg_bytes_per_launch": algb * batch / max(launches, 1),
            "avg_launch_us": g_avg * 1e3 / max(launches, 1),
            "hbm_counter_GBs": (traffic / (g_avg * 1e-3) / 1e9) if traffic else None,  # what HBM really carried
            "hbm_counter_frac": (traffic / (g_avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "limiter": "LDS array + VALU issue of the T-table AES core (see aes_core); HBM is at hbm_counter_frac",
        },
        # what actually limits the kernels: AES blocks through the LDS T-table core (VALU issue + LDS address path)
        "aes_core": {
            "garble_blocks_per_s": blocks_g * batch / (g_avg * 1e-3),
            "eval_blocks_per_s": blocks_e * batch / (e_avg * 1e-3),
            "peak_blocks_per_s": AES_CORE_PEAK_BLOCKS,
            "frac_garble": blocks_g * batch / (g_avg * 1e-3) / AES_CORE_PEAK_BLOCKS,
            "frac_eval": blocks_e * batch / (e_avg * 1e-3) / AES_CORE_PEAK_BLOCKS,
            "peak_source": "tools/aes_ubench: the production AES core alone, 16 waves/CU (own micro-benchmark)",
            # against the hardware figure instead: table look-ups/s over the LDS array's ds_read_b32 rate
            "lds_array_frac_garble": lds_g,
            "lds_array_frac_eval": lds_e,
            "lds_array_peak_lookups_per_s": LDS_B32_PEAK_LOOKUPS,
        },
    }
    if collective:
        # which GPU every rank really ran on (one process per GPU: each may see its own as device 0), as the ranks say
        # themselves: 96 bytes per rank through the communicator's own all-gather
        me = ("%s|%s|%d" % (ctx.pci_bus_id() if hasattr(ctx, "pci_bus_id") else "?", os.environ.get("HIP_VISIBLE_DEVICES", ""),
                            ctx.device)).encode()[:95]
        cards = comm.allgather_host(np.frombuffer(me.ljust(96, b"\0"), np.uint8))
        rank_devices = []
        for r in range(world):
            bus, vis, dev = (bytes(cards[r]).rstrip(b"\0").decode(errors="replace").split("|") + ["", "", ""])[:3]
            rank_devices.append({"rank": r, "pci_bus_id": bus, "HIP_VISIBLE_DEVICES": vis or None, "device_index": dev})
        res["config4"] = {
            "rank_devices": rank_devices,
            "distinct_gpus": len({d["pci_bus_id"] for d in rank_devices}),
            "workload": "aes_128 x %d instances = %d per GPU x %d GPUs (BASELINE config 4 is 8 192 x 8), outputs of %d steps per "
                        "gather" % (batch * world, batch, world, K),
            "instances_total": batch * world,
            "gathered_bytes_per_gpu": int(d_acc.nbytes),
            "gathered_bytes_total": int(d_acc.nbytes) * world,
            "gather_us": gather_us,
            "gather_GBs_per_gpu_received": (int(d_acc.nbytes) * (world - 1) / (gather_us * 1e-6) / 1e9) if gather_us and world > 1 else None,
            "gathers_in_timed_region": loop.gathers,
            # (GC_RCCL_PATH: another library under gc_comm_* — tests/standin_rccl puts N ranks on one GPU; its times mean nothing)
            "collective": ("ncclAllGather (gc_comm_allgather, RCCL over xGMI), one per %d steps" % K) if not os.environ.get("GC_RCCL_PATH")
                          else "gc_comm_allgather over the library of GC_RCCL_PATH (%s) — NOT RCCL, one per %d steps"
                               % (os.path.basename(os.environ["GC_RCCL_PATH"]), K),
            "gathered_outputs_ok": ok,
        }
    if comm is not None:
        # every rank's own time over the same K steps (the job's is the slowest rank's) -> per-rank AND/s, min / max over ranks
        mine = np.array([getattr(loop, "local_elapsed", elapsed)], np.float64)
        times = np.frombuffer(np.ascontiguousarray(comm.allgather_host(mine.view(np.uint8))).tobytes(), np.float64)
        vals = [n_and * batch * args.steps / t for t in times]
        res["per_rank"] = {"elapsed_s": [float(t) for t in times], "value": vals, "min": min(vals), "max": max(vals),
                           "unit": "AND-gates/s per GPU over the timed steps (the job's value is all ranks' units over the slowest rank's time)"}
    gb.close()
    ev.close()
    dc.close()
    for d in (d_rnd, d_bits, d_acc, d_mis, d_all):
        if d is not None:
            d.close()
    if (world > 1 and comm is not None and not args.no_synthetic and args.circuit.endswith("aes_128.gcf")
            and not os.environ.get("GC_BENCH_ENGINE")):
        # north star: "AND-gates/sec on synthetic levelised circuits reported at 1/2/4/8 GPUs": every rank runs the five rows
        # of the N = 1 line on its own GPU (independent instances, 1 024 per GPU as there), the job's figure is the sum
        stage[0] = "synthetic rows"
        rows = sweep_rows_for_line(1024, key, ctx)
        mine = np.array([[r.get("and_gates_per_s") or 0.0, r.get("gates_per_s") or 0.0] for r in rows], np.float64)
        allv = np.frombuffer(np.ascontiguousarray(comm.allgather_host(mine.reshape(-1).view(np.uint8))).tobytes(),
                             np.float64).reshape(world, len(rows), 2)
        for i, r in enumerate(rows):
            r["and_gates_per_s_job"] = float(allv[:, i, 0].sum())
            r["gates_per_s_job"] = float(allv[:, i, 1].sum())
            r["and_gates_per_s_per_rank"] = [float(v) for v in allv[:, i, 0]]
            r["instances_per_gpu"] = 1024
        res["synthetic"] = rows
    if rank == 0 and world == 1:
        aes = args.circuit.endswith("aes_128.gcf")
        # how long each group of side rows took (the timed region of `value` is ms_per_step x steps; the rest of the run is these)
        side_wall = {}
        t_side = [time.perf_counter()]

        def lap(name):
            now = time.perf_counter()
            side_wall[name] = round(now - t_side[0], 2)
            t_side[0] = now
        if not args.no_synthetic and aes:
            # SURVEY §8d / north star: synthetic levelised circuits as absolute numbers and as fractions of the rooflines
            res["synthetic"] = sweep_rows_for_line(batch, key, ctx)
            lap("synthetic")
        if not args.no_extra_rows and aes:
            res.update(reference_bench_rows(batch, circ, ctx))
            res["level_launch"] = level_launch_row(batch, circ, key, ctx)
            res["n1_batch8192"] = per_gpu_share_row(8192, circ, key, ctx)
            lap("key16 + and_chain + level_launch")
        if not args.no_iknp:
            # second kernel pair of the path (ot/iknp.go) and its callers (COT pads over MITCCRH, KOS check, bit-COT):
            # device-resident API, 4 Mi OTs
            from scripts.bench_ot import run as ot_run
            ot = ot_run(1 << 22, 5, ctx=ctx)
            res["iknp"] = ot.pop("iknp")
            res["ot"] = ot
            lap("ot")
        if not args.no_host_api and aes:
            # the literal drop-in calls with HOST buffers (PCIe-inclusive; never `value`), see DESIGN.md §7
            from scripts.bench_host_api import run as host_api_run
            res["host_api"] = host_api_run(batch, 8, key)
            lap("host_api")
        if not args.no_stream and aes:
            # config 5 shape: ONE instance through gc_stream_* (garbler pipelined, evaluator over the produced bytes);
            # bounded samples of scripts/bench_stream.py, SHA-256 of the byte streams checked against the oracle-made
            # goldens (tests/golden/stream_bench_golden.json) inside
            from scripts.bench_stream import run_for_line as stream_run
            res["stream"] = stream_run(key=key, ctx=ctx)
            lap("stream")
        if not args.no_config3 and aes:
            # config 3: sha256xor x 256, the evaluator's labels through the 65 536-OT IKNP extension + COT pads, all on the
            # device; every digest checked against hashlib inside (scripts/bench_config3.py)
            from scripts.bench_config3 import run as config3_run
            res["config3"] = config3_run(256, 10, key)
            lap("config3")
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(circ, key)
            lap("cpu_baseline")
        # "probe `go version` first" — with a Go toolchain on the box the baseline would be the reference itself
        import shutil
        go = shutil.which("go")
        go_ver = None
        if go:
            try:
                import subprocess
                go_ver = subprocess.run([go, "version"], capture_output=True, text=True, timeout=20).stdout.strip()
            except Exception as e:  # noqa: BLE001
                go_ver = "go found, `go version` failed: %s" % e
        res.setdefault("cpu_baseline", {})["go_on_box"] = go_ver or False
        res["side_rows_wall_s"] = side_wall
    stage[0] = "shutdown"
    if comm is not None:
        comm.barrier()
        comm.close()
    ctx.close()
    if dog:
        dog.disarm()
    if rank == 0:
        # the JSON line goes out LAST: flush whatever native libraries (RCCL banner) left in C stdio first
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
