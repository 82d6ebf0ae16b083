#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_6; mkdir -p $OUT
python -m pytest tests/test_gpu_garble_eval.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.log
python scripts/bench_host_api.py 1024 > $OUT/host_api.json 2> $OUT/host_api.err; echo "host api rc=$?"; cat $OUT/host_api.json; tail -3 $OUT/host_api.err
GC_TRACE=1 python - <<'P' 2>&1 | tail -24
import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
from scripts.bench_host_api import Party, p
from mpc_amd import engine, parse_file
c = parse_file('tests/golden/aes_128.gcf')
L = engine.lib(); batch=1024
k = np.frombuffer(bytes(range(32)), np.uint8).copy()
rnd = np.frombuffer(np.random.default_rng(1).bytes(batch * 16 * (c.num_inputs + 1)), np.uint8).copy()
P = Party(c, batch, True)
for r in range(3):
    print("--- rep", r, file=sys.stderr)
    assert L.gc_garble(P.dc.h, p(k), len(k), p(rnd), len(rnd), batch, p(P.R), None, p(P.io), p(P.slabs[0])) == 0
    P.inputs[...] = P.io[:, :c.num_inputs]["l0"]
    assert L.gc_eval(P.dc.h, p(k), len(k), batch, None, p(P.inputs), p(P.slabs[0]), P.rows, p(P.outl)) == 0
P.close()
P
