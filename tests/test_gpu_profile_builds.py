"""The instrumented builds of the fused kernels (gc_batch_debug_profile: s_memtime marks around the phases of a unit /
a level, a developer aid) are separate template instantiations of the production code: they must produce the same
tables and labels as the plain builds, on every kernel family."""
import numpy as np
import pytest

from mpc_amd import engine
from mpc_amd.circuit import synthetic_levelised

pytestmark = pytest.mark.gpu
KEY = bytes(range(32))


def run_pair(ctx, c, batch, schedule=1, key=KEY):
    dc = engine.DeviceCircuit(ctx, c)
    res = []
    for prof in (False, True):
        gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
        if schedule != 1:
            gb.set_schedule(schedule); ev.set_schedule(schedule)
        gen = np.random.default_rng(99)
        d_rnd = ctx.to_device(gen.integers(0, 256, (batch, c.num_inputs + 1, 16), dtype=np.uint8))
        d_bits = ctx.to_device(gen.integers(0, 2, (batch, c.num_inputs), dtype=np.uint8))
        d_out = ctx.zeros((batch, c.num_outputs))
        d_mis = ctx.zeros(1, np.int32)
        if prof:
            gb.debug_profile(True); ev.debug_profile(True)
        gb.garble(key, d_rnd); ev.select_inputs(gb, d_bits); ev.eval(key, gb)
        gb.decode(ev, d_out, d_mis)
        ctx.sync()
        assert int(d_mis.numpy()[0]) == 0
        out, bits = d_out.numpy(), d_bits.numpy()
        for i in (0, batch - 1):
            assert (c.compute_bits(bits[i])[c.NumWires - c.num_outputs:] == out[i]).all()
        if prof:
            for b in (gb, ev):
                assert b.debug_profile(True, read=True).sum() > 0
        res.append(out)
        gb.close(); ev.close()
    assert (res[0] == res[1]).all()
    dc.close()


def test_instrumented_flat_kernels(aes_circ, sha_circ):
    ctx = engine.Context(0)
    run_pair(ctx, aes_circ, 64)   # wide units with column-sliced tails
    run_pair(ctx, sha_circ, 5)    # narrow units
    ctx.close()


def test_instrumented_level_walking_kernels(aes_circ):
    ctx = engine.Context(0)
    run_pair(ctx, aes_circ, 40, schedule=2)
    ctx.close()


@pytest.mark.parametrize("shape", [(128, 1024, 0.17), (14, 2500, 0.3), (900, 64, 0.3)])
def test_instrumented_hbm_wire_kernels(shape):
    """levels that mix hash lanes and XOR lanes over several passes (the instrumented build of the grouped passes
    faulted here in round 2), wide levels with every gate type, and single-pass levels"""
    L, W, f = shape
    ctx = engine.Context(0)
    c = synthetic_levelised(L, W, f, seed=105, ninputs=256, or_frac=0.03 if W == 2500 else 0.0,
                            inv_frac=0.05 if W == 2500 else 0.0, xnor_frac=0.1 if W == 2500 else 0.0)
    dc = engine.DeviceCircuit(ctx, c)
    b = engine.Batch(dc, 64)
    assert not b.lds_wires
    b.close(); dc.close()
    run_pair(ctx, c, 64)
    if W == 1024:  # the AES-128 / AES-192 instantiations of the same kernels
        run_pair(ctx, c, 64, key=KEY[:16])
        run_pair(ctx, c, 64, key=KEY[:24])
    ctx.close()
