"""GC_STREAM_DATAFLOW=1 (round 6, an experiment, off by default): the garbler's launch units ordered ACROSS launches by per-wire version
and reader counts on the device instead of by stream order (mpc_amd/csrc/stream_internal.h: Dataflow; kernels.h: DfBlock) — groups
on rotating streams, deep steps on their lanes without events towards the groups, big steps joining everything and bumping the
counts themselves.  The bytes on the wire and every wire label are the serial loop's whatever runs beside what: the same
programs as the default path, against the oracle.

Reference: circuit/stream_garble.go:131-157 (Get / Set through in[] / out[]), :161-192."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["1", "3", "4"], ids=["versions", "versions+pool", "versions+persistent-pool"])
def _dataflow(monkeypatch, request):
    """1: units wait for their wires' versions, one launch's workgroups run that launch's units; 3: out-of-order issue on top —
    every unit is published into a ring and the workgroups of ANY launch claim the lowest unclaimed one (kernels.h: PoolCtl); 4: the
    same with PERSISTENT workgroups that wait for publications"""
    monkeypatch.setenv("GC_STREAM_DATAFLOW", request.param)
    if request.param == "4" and int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 16:
        pytest.skip("persistent workgroups need GPU_MAX_HW_QUEUES >= 16, set before the process's first HIP call "
                    "(GPU_MAX_HW_QUEUES=16 python -m pytest tests/test_gpu_dataflow.py -m gpu)")


def test_ed25519like_under_dataflow():
    from tests.test_gpu_stream import test_stream_ed25519like_matches_oracle
    test_stream_ed25519like_matches_oracle()


def test_instruction_mix_under_dataflow(monkeypatch):
    """(deep steps on lanes, groups on rotating streams: no event between them)"""
    from tests.test_gpu_stream import test_stream_instruction_mix_matches_oracle
    test_stream_instruction_mix_matches_oracle(False, monkeypatch)


def test_mixed_program_with_big_steps_under_dataflow(monkeypatch):
    """(big steps on the ctx stream join every group launched so far and bump the wires' counts themselves)"""
    from tests.test_gpu_stream import test_stream_mixed_program_matches_oracle
    test_stream_mixed_program_matches_oracle(False, monkeypatch)


@pytest.mark.parametrize("base,keylen,window,by_handle", [(0, 32, 300, True), (0x20000, 16, 300, False), (0xfff0, 24, 9, True)])
def test_fused_chains_under_dataflow(base, keylen, window, by_handle):
    from tests.test_gpu_stream_fuse import test_fused_chains_match_oracle
    test_fused_chains_match_oracle(base, keylen, window, by_handle)


def test_chain_cut_by_read_backs_under_dataflow():
    """(gc_stream_get_wire / set_wire in the middle of a program: the read-back drains the rotating streams, an upload of host-set
    labels goes behind every group launched so far)"""
    from tests.test_gpu_stream_fuse import test_fused_chain_cut_by_a_read_back_and_by_set_wire
    test_fused_chain_cut_by_a_read_back_and_by_set_wire()


def test_in_place_updates_and_aliases_under_dataflow():
    from tests.test_gpu_stream import test_stream_in_out_alias_the_same_global_wire, test_stream_outputs_that_are_input_wires
    test_stream_in_out_alias_the_same_global_wire()
    test_stream_outputs_that_are_input_wires()


@pytest.mark.parametrize("seed", [0, 3, 14, 22, 41, 75, 96, 150, 301, 302, 303, 304, 305, 306])
def test_scheduling_fuzz_under_dataflow(seed):
    """tests/queued_case.py: random programs queued 1 to 300 steps ahead — overwritten live wires (WAR / WAW across launches),
    repeated operands, in-place updates, 0 to 3 lanes, big steps"""
    from tests.queued_case import run_case
    run_case(seed)


def test_c_host_under_dataflow():
    import os
    from scripts import bench_stream as bs
    if not os.path.exists(bs.NATIVE):
        pytest.skip("tools/stream_driver is not built")
    r = bs.run_native("ed25519like1", bytes(range(32)), 1024)  # (the child inherits GC_STREAM_DATAFLOW)
    assert r["sha256_ok"] is True
