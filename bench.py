#!/usr/bin/env python3
"""bench.py — AND-gates/s (garble+eval) of the AES-128 circuit batch on MI355X.

One "step" = one pass of the hot path over one batch of synthetic instances that are already
resident in HBM: Circuit.Garble for `batch` instances, hand-over of the evaluator's input labels,
Circuit.Eval, and BitFromLabel decoding (+ the RCCL gather of the decoded outputs when N > 1).
Workload at N=1: BASELINE.json configs[1] — aes_128 (36 663 gates, 6 400 AND) x 1 024 instances,
32-byte garbling key (AES-256, as circuit.Garbler uses).  N > 1: the same per-GPU batch on every
rank (weak scaling, independent instances, no data-path collective except the output gather).

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
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
# Secondary bound (SURVEY.md §8d "LDS bandwidth / VALU issue on AND-dense levels"): the fixed-key hash is a
# T-table AES out of LDS.  tools/aes_ubench measures the production AES core alone at 10.28 cycles per block per
# CU with 16 waves/CU (59.7 G blocks/s on 256 CUs; profiles/r01_aes_ubench.txt).
AES_CORE_PEAK_BLOCKS = 59.7e9
AES_BLOCKS = {"and": (4, 2), "inv": (2, 1), "or": (4, 1)}  # (garble, eval) distinct AES blocks per gate


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--circuit", default=os.path.join(ROOT, "tests", "golden", "aes_128.gcf"))
    ap.add_argument("--key-bytes", type=int, default=32, choices=[16, 24, 32])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-iknp", action="store_true", help="skip the IKNP OT-extension side measurement")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--force-collective", action="store_true", help="run the output gather even with one rank (testing)")
    ap.add_argument("--schedule", type=int, default=1)
    ap.add_argument("--check", action="store_true", help="verify decoded outputs against plaintext evaluation")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py --gpus %d must be launched through torch.distributed.run" % args.gpus, file=sys.stderr)
            sys.exit(2)
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    from mpc_amd import engine, parse_file

    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_collective:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:  # --force-collective without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    circ = parse_file(args.circuit)
    circ.name = os.path.splitext(os.path.basename(args.circuit))[0]
    key = bytes(range(args.key_bytes))
    ctx = engine.Context(local_rank)
    dc = engine.DeviceCircuit(ctx, circ)
    info = dc.info
    batch = args.batch
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    for b in (gb, ev):
        b.set_graph(not args.no_graph)
        b.set_schedule(args.schedule)

    # synthetic inputs, resident in HBM before the timed region
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)
    d_rnd = torch.randint(0, 256, (batch, circ.num_inputs + 1, 16), dtype=torch.uint8, device="cuda", generator=gen)
    d_bits = torch.randint(0, 2, (batch, circ.num_inputs), dtype=torch.uint8, device="cuda", generator=gen)
    collective = world > 1 or args.force_collective
    # Decoded outputs of GATHER_EVERY steps are collected in one accumulator and gathered with ONE all_gather (fewer,
    # larger collectives: a gather per step costs ~3 % even overlapped — its kernel shares the CUs with the next garble);
    # two accumulators, so that the gather of one overlaps the steps that fill the other.
    K = GATHER_EVERY if collective else 1
    nacc = 2 if collective else 1
    d_acc = [torch.zeros((K, batch, circ.num_outputs), dtype=torch.uint8, device="cuda") for _ in range(nacc)]
    d_out = d_acc[0][0]
    d_mis = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_all = [torch.zeros((world, K, batch, circ.num_outputs), dtype=torch.uint8, device="cuda") for _ in range(nacc)] \
        if collective else None

    def device_step(a=0, j=0):
        gb.garble(key, d_rnd.data_ptr())
        ev.select_inputs(gb, d_bits.data_ptr())
        ev.eval(key, gb)
        gb.decode(ev, d_acc[a][j].data_ptr(), d_mis.data_ptr())

    graphs = None  # the step's kernels recorded once per output slot in a hipGraph (gc_ctx_capture_*): one launch per step
    # The only collective: all_gather of the decoded outputs over RCCL/xGMI, on torch's stream.  The engine runs on
    # its own HIP stream; the two are chained with events (no host synchronisation inside the loop): the gather of an
    # accumulator waits for the decode that filled its last slot and runs while the next steps fill the other one; the
    # first decode into an accumulator waits for its previous gather.
    eng_stream = torch.cuda.ExternalStream(ctx.stream) if collective else None
    done_ev = [torch.cuda.Event() for _ in range(nacc)]
    gathered_ev = [None] * nacc
    counter = [0]

    def gather(a):
        done_ev[a].record(eng_stream)
        torch.cuda.current_stream().wait_event(done_ev[a])
        dist.all_gather_into_tensor(d_all[a].view(world * K * batch, -1), d_acc[a].view(K * batch, -1))
        e = torch.cuda.Event()
        e.record()
        gathered_ev[a] = e

    def step():
        i = counter[0]
        counter[0] += 1
        a, j = (i // K) % nacc, i % K
        if collective and j == 0 and gathered_ev[a] is not None:
            eng_stream.wait_event(gathered_ev[a])
        if graphs is not None:
            graphs[a][j].launch()
        else:
            device_step(a, j)
        if collective and j == K - 1:
            gather(a)

    def flush():  # outputs of the steps since the last full accumulator
        if collective and counter[0] % K:
            gather((counter[0] // K) % nacc)
            counter[0] += K - counter[0] % K

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    torch.cuda.synchronize()  # inputs were generated on torch's stream; the engine runs on its own
    device_step()  # first call uploads the round keys (not capturable), and warms the allocator
    ctx.sync()
    if not args.no_graph and args.schedule != 0:
        try:
            graphs = [[ctx.capture(lambda a=a, j=j: device_step(a, j)) for j in range(K)] for a in range(nacc)]
        except Exception as e:  # capture is an optimisation: fall back to direct launches of the same kernels
            print("bench: hipGraph capture unavailable (%s); launching directly" % e, file=sys.stderr)
            graphs = None
            ctx.sync()
    for _ in range(args.warmup):
        step()
    flush()
    fence()
    g_ms, e_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    flush()  # every output of the timed steps is gathered inside the timed region
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-pass device times (events on the engine stream) from a few extra, untimed-by-wall steps
    for _ in range(min(20, max(1, args.steps))):
        gb.garble(key, d_rnd.data_ptr())
        g_ms.append(gb.last_ms)
        ev.select_inputs(gb, d_bits.data_ptr())
        ev.eval(key, gb)
        e_ms.append(ev.last_ms)
    ctx.sync()
    mismatches = int(d_mis.cpu()[0])

    ok = mismatches == 0
    if collective:  # the gathered tensor holds this rank's outputs at its offset
        for a in range(nacc):
            ok = ok and bool(torch.equal(d_all[a][rank], d_acc[a]))
    if args.check:
        bits = d_bits.cpu().numpy()
        out = d_out.cpu().numpy()
        for i in range(0, batch, max(1, batch // 16)):  # against plaintext evaluation (circuit/computer.go)
            plain = circ.compute_bits(bits[i])
            ok = ok and bool((plain[circ.NumWires - circ.num_outputs:] == out[i]).all())

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
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "latest_pmc.json")
    if os.path.exists(pmc):
        try:
            with open(pmc) as f:
                pj = json.load(f)
            if pj.get("batch") == batch and pj.get("schedule") == args.schedule and pj.get("key_bytes") == args.key_bytes:
                traffic = pj.get("garble_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    res = {
        "metric": "AND-gates/sec (garble+eval), AES-128 circuit batch",
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
        },
        "garble_ms": g_avg,
        "eval_ms": e_avg,
        "and_gates_per_s_garble_only": n_and * batch / (g_avg * 1e-3),
        "and_gates_per_s_eval_only": n_and * batch / (e_avg * 1e-3),
        "hbm_alg_GBs_garble_plus_eval": 2 * algb * batch / ((g_avg + e_avg) * 1e-3) / 1e9,
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "alg_bytes_per_launch": algb * batch / max(launches, 1),
            "avg_launch_us": g_avg * 1e3 / max(launches, 1),
        },
        # what actually limits the kernels: AES blocks through the LDS T-table core (VALU issue + LDS address path)
        "aes_core": {
            "garble_blocks_per_s": blocks_g * batch / (g_avg * 1e-3),
            "eval_blocks_per_s": blocks_e * batch / (e_avg * 1e-3),
            "peak_blocks_per_s": AES_CORE_PEAK_BLOCKS,
            "frac_garble": blocks_g * batch / (g_avg * 1e-3) / AES_CORE_PEAK_BLOCKS,
            "frac_eval": blocks_e * batch / (e_avg * 1e-3) / AES_CORE_PEAK_BLOCKS,
        },
    }
    if rank == 0:
        if world == 1 and not args.no_iknp:
            # second kernel pair of the path (ot/iknp.go): OT extension on the device-resident API, 4 Mi OTs
            from scripts.bench_iknp import run as iknp_run
            res["iknp"] = iknp_run(1 << 22, 5, ctx=ctx)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(circ, key)
    gb.close()
    ev.close()
    dc.close()
    ctx.close()
    if collective:
        dist.destroy_process_group()
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
