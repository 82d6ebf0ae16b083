"""BASELINE config 4 on the GPU: aes_128 x 65 536 instances sharded 8 192 per GPU, RCCL gather of the outputs.

One MI355X runs ONE rank's share at full size (8 192 instances, device-resident API): decoded outputs == AES-128 for
every instance (size-independent property), byte parity with the oracle on sampled instances, and the rank's outputs
travel through the same gc_comm all-gather the N-rank job uses (a one-rank communicator: RCCL itself is exercised).
With two or more devices in the box the sharded job itself runs: one gc_ctx per device, gc_comm_init_all, shards of
512 instances, gathered bits checked against plaintext AES and the oracle (circuit/garble.go:253-278:
instances are independent)."""
import numpy as np
import pytest

import oracle
from mpc_amd import engine
from mpc_amd.dist import shard_range
from tests.util import bits_lsb, drbg, int_from_bits

pytestmark = pytest.mark.gpu

KEY256 = bytes(range(32))
PER_GPU = 8192


def aes_inputs(lo, hi):
    """instance i of the global batch: AES key / plaintext from the DRBG, as circuit input bits"""
    keys = [drbg("c4k%d" % i, 16) for i in range(lo, hi)]
    pts = [drbg("c4p%d" % i, 16) for i in range(lo, hi)]
    bits = np.zeros((hi - lo, 256), np.uint8)
    for j in range(hi - lo):
        bits[j, :128] = bits_lsb(int.from_bytes(keys[j], "big"), 128)
        bits[j, 128:] = bits_lsb(int.from_bytes(pts[j], "big"), 128)
    return keys, pts, bits


def run_shard(ctx, dc, schedule, rnd, bits):
    """garble -> input hand-over -> eval -> decode of one shard; returns (garbler batch, decoded bits buffer).  Every
    device buffer comes from the C ABI (gc_dev_alloc on the shard's own ctx / device): no torch in the process."""
    batch = len(bits)
    gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
    gb.set_schedule(schedule)
    ev.set_schedule(schedule)
    d_rnd = ctx.to_device(rnd)
    d_bits = ctx.to_device(bits)
    d_out = ctx.zeros((batch, 128))
    d_mis = ctx.zeros(1, np.int32)
    gb.garble(KEY256, d_rnd)
    ev.select_inputs(gb, d_bits)
    ev.eval(KEY256, gb)
    gb.decode(ev, d_out, d_mis)
    ctx.sync()
    assert int(d_mis.numpy()[0]) == 0
    ev.close()
    return gb, d_out, (d_rnd, d_bits)


@pytest.mark.parametrize("schedule", [1, 0])
def test_aes128_x8192_one_rank_share(aes_circ, schedule):
    c = aes_circ
    ctx = engine.Context(0)
    dc = engine.DeviceCircuit(ctx, c)
    stride = 16 * (c.num_inputs + 1)
    rnd = drbg("c4rnd", stride * PER_GPU)
    keys, pts, bits = aes_inputs(0, PER_GPU)
    gb, d_out, keep = run_shard(ctx, dc, schedule, rnd, bits)
    assert gb.tile_instances == (4 if schedule == 1 else 1)
    out = d_out.numpy()
    for i in range(PER_GPU):  # every instance: decoded ciphertext == AES-128(key, pt)
        assert int_from_bits(out[i]).to_bytes(16, "big") == oracle.aes_encrypt(keys[i], pts[i]), "instance %d" % i
    # byte parity with the oracle on 16 sampled instances: R, every table row, the output wires' zero labels
    R = gb.read_r()
    outl = gb.read_outputs()
    slab = gb.read_slab()
    for i in [0, 1, 3, 4, 63, 64, 1023, 1024, 2047, 4095, 4096, 5000, 6143, 8000, 8190, 8191]:
        ref = oracle.garble(c.Gates, c.NumWires, c.num_inputs, KEY256, rnd[i * stride:(i + 1) * stride])
        assert R[i] == ref["R"], "R of instance %d" % i
        assert (slab[i] == ref["slab"]).all(), "slab of instance %d" % i
        assert (outl[i] == ref["wires"]["l0"][c.NumWires - c.num_outputs:]).all(), "output labels of instance %d" % i
    del slab
    # the terminal exchange: this rank's decoded bits through gc_comm_allgather (one-rank communicator)
    if schedule == 1:
        comm = engine.Comm(ctx, engine.comm_unique_id(), 1, 0)
        d_all = ctx.zeros(d_out.shape)
        comm.allgather(d_out, d_all, d_out.nbytes)
        comm.barrier()
        assert (d_all.numpy() == out).all()
        assert comm.allreduce_max(3.25) == 3.25
        comm.close()
    gb.close(); dc.close(); ctx.close()


def test_two_devices_sharded_gather(aes_circ):
    """the N-rank job in one process: gc_comm_init_all over the box's devices, contiguous shards, one all-gather"""
    ndev = min(engine.device_count(), 8)
    if ndev < 2:
        pytest.skip("one device in this box (the driver's multi-GPU node runs it)")
    c = aes_circ
    per = 512
    total = per * ndev
    stride = 16 * (c.num_inputs + 1)
    rnd = drbg("c4multi", stride * total)
    keys, pts, bits = aes_inputs(0, total)
    ctxs = [engine.Context(d) for d in range(ndev)]
    comms = engine.Comm.init_all(ctxs)
    dcs, gbs, outs, alls, keep = [], [], [], [], []
    for d in range(ndev):
        lo, hi = shard_range(total, d, ndev)
        assert hi - lo == per
        dcs.append(engine.DeviceCircuit(ctxs[d], c))
        gb, d_out, k = run_shard(ctxs[d], dcs[d], 1, rnd[lo * stride:hi * stride], bits[lo:hi])
        gbs.append(gb); outs.append(d_out); keep.append(k)
        alls.append(ctxs[d].zeros((ndev, per, 128)))
    engine.Comm.allgather_all(comms, outs, alls, per * 128)
    for cm in comms:
        cm.ctx.sync()
    ref = alls[0].numpy().reshape(total, 128)
    for d in range(1, ndev):  # every rank holds the whole result
        assert (alls[d].numpy().reshape(total, 128) == ref).all()
    for i in range(total):
        assert int_from_bits(ref[i]).to_bytes(16, "big") == oracle.aes_encrypt(keys[i], pts[i]), "instance %d" % i
    for d in range(ndev):  # shard boundaries against the oracle
        lo, hi = shard_range(total, d, ndev)
        R, slab = gbs[d].read_r(), gbs[d].read_slab()
        for i in (lo, hi - 1):
            o = oracle.garble(c.Gates, c.NumWires, c.num_inputs, KEY256, rnd[i * stride:(i + 1) * stride])
            assert R[i - lo] == o["R"] and (slab[i - lo] == o["slab"]).all()
    for d in range(ndev):
        comms[d].close(); gbs[d].close(); dcs[d].close(); ctxs[d].close()
