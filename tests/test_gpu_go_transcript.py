"""The reference's TestDeterministicTranscript (sha2pc/sha2pc_test.go:73-130) with the HIP engine as `Circuit.Garble`: the
SHA-256 of the encoded round 3 — garbling key, all 42 914 table labels of sha256xor.mpclc, input labels, output wires, OT
ciphertexts — must be the constants the Go tests hold (`expRound3`, :123; `idemRound3Hash`, :415).  No oracle in between: GPU bytes against Go's."""
import pytest

from mpc_amd import engine

import go_transcript as gt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case,schedule", [("transcript", None), ("transcript", 0), ("transcript", 1), ("transcript", 2),
                                           ("idempotency", None)])
def test_round3_bytes_equal_the_go_constant(ctx, sha_circ, case, schedule):
    dc = engine.DeviceCircuit(ctx, sha_circ, schedule=schedule)

    def garble(key, rnd):
        g = dc.garble(key, rnd, batch=1)  # gc_garble: the C ABI under Circuit.Garble (include/gcengine.h)
        io = g["io"][0]
        return {"in": io[:512], "out": io[512:]}, g["slab"][0]

    t = gt.transcript(sha_circ, garble, case)
    want = gt.CASES[case][1]
    assert (t["round1"], t["round2"]) == want[:2]
    assert len(t["round3_bytes"]) == gt.ROUND3_LEN
    assert t["round3"] == want[2]
    # EvaluatorRound4 with gc_eval as Circuit.Eval: the digest the Go tests expect (`expFinal`, :124)
    assert gt.evaluator_round4(t, lambda key, slab, inputs: dc.eval(key, slab, inputs=inputs[None, :], batch=1)[0]) == gt.EXP_FINAL
    dc.close()


def test_round3_from_the_device_pipeline(ctx, sha_circ):
    """The same constant through the device-resident calls: gc_batch_garble on a device random stream, the table section of
    round 3 written by gc_batch_egress_tables_dense (sha2pc's encodeGarbledTables, encoding.go:363-411, as a kernel)."""
    import numpy as np
    from oracle import LABEL
    c = sha_circ
    dc = engine.DeviceCircuit(ctx, c)
    gb = engine.Batch(dc, 1)
    gb.set_store_all(True)
    nbytes = 16 * 42914
    seen = {}

    def garble(key, rnd):
        gb.garble(key, ctx.to_device(rnd))
        d_out = ctx.zeros(nbytes)
        gb.egress_tables_dense(d_out, nbytes)
        raw = d_out.numpy().tobytes()
        seen["raw"] = raw
        be = np.frombuffer(raw, ">u8").reshape(-1, 2)
        slab = np.zeros(len(be), LABEL)
        slab["d0"], slab["d1"] = be[:, 0], be[:, 1]
        w = gb.read_wires()[0]
        return {"in": w[:512], "out": w[c.NumWires - 256:]}, slab

    t = gt.transcript(c, garble, "transcript")
    assert t["round3_bytes"][42:42 + nbytes] == seen["raw"]  # (the device's bytes are what was hashed)
    assert t["round3"] == gt.CASES["transcript"][1][2]
    gb.close(); dc.close()
