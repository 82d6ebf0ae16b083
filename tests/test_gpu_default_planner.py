"""The product's DEFAULT chain-fusion path under the parity tests: tests/conftest.py sets GC_STREAM_FUSE_EAGER for
the whole suite — merged plans at first sight, on the calling thread — so that every chain runs on its fused plan
deterministically.  A product run has no such variable: a chain met for the first times runs its steps one after the other in
ONE workgroup through the wire store (k_*_flat_jobs<.., CHAIN>, pad_ / pad2_ records, the AES table kept from the first job),
the ctx's planner thread builds the merged plan at the third sighting, and later sightings run fused.  Here the variable is
unset and the same programs — the Ed25519-shaped one, the 23-circuit instruction mix, the scheduling fuzz — run against the
oracle: planned and unplanned units in one run.  (bench.py runs outside pytest, i.e. on this default path too, and refuses a
stream whose SHA-256 is not the oracle's.)

Reference: circuit/stream_garble.go:161-192, compiler/ssa/streamer.go:412-524 (one circuit per SSA instruction)."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _default_planner(monkeypatch):
    monkeypatch.delenv("GC_STREAM_FUSE_EAGER", raising=False)


def test_ed25519like_on_the_default_planner():
    from tests.test_gpu_stream import test_stream_ed25519like_matches_oracle
    test_stream_ed25519like_matches_oracle()


def test_instruction_mix_on_the_default_planner(monkeypatch):
    from tests.test_gpu_stream import test_stream_instruction_mix_matches_oracle
    test_stream_instruction_mix_matches_oracle(False, monkeypatch)


@pytest.mark.parametrize("base,keylen,window,by_handle", [(0, 32, 300, True), (0xfff0, 24, 9, False)])
def test_fused_chains_on_the_default_planner(base, keylen, window, by_handle):
    """the chain programs of tests/test_gpu_stream_fuse.py three times over one ctx: first sightings unplanned, the planner thread
    catches up, later sightings fused — every byte and label the oracle's each time"""
    import time
    from mpc_amd import engine
    from tests.test_gpu_stream_fuse import _chain_program, _run
    from tests.util import drbg
    ctx = engine.Context(0)
    steps, prim = _chain_program(base)
    key, rnd = drbg("dp%d" % base, keylen), drbg("dp-r%d" % base, 16 * (len(prim) + 1))
    fused = []
    for rep in range(3):
        _, gf, ef = _run(ctx, steps, prim, key, rnd, window, by_handle=by_handle)
        fused.append((gf, ef))
        time.sleep(0.2)
    ctx.close()


@pytest.mark.parametrize("seed", [0, 3, 14, 22, 41, 75, 96, 150, 201, 202, 203, 204])
def test_scheduling_fuzz_on_the_default_planner(seed):
    """tests/queued_case.py (ext_fuzz's seventh pass: random programs queued 1 to 300 steps ahead over groups, deep steps, lanes,
    big steps, overwritten wires) with plans made in the background"""
    from tests.queued_case import run_case
    run_case(seed)


def test_c_host_ed25519like_on_the_default_planner():
    """the C host (a child process: no pytest environment reaches it unless handed over) without the variable"""
    from scripts import bench_stream as bs
    if not os.path.exists(bs.NATIVE):
        pytest.skip("tools/stream_driver is not built")
    assert "GC_STREAM_FUSE_EAGER" not in os.environ
    r = bs.run_native("ed25519like1", bytes(range(32)), 1024)
    assert r["sha256_ok"] is True
