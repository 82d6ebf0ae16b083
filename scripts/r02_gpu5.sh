#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
GC_TRACE=1 python - <<'P' 2>&1 | tail -60
import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
from scripts.bench_host_api import Party, p
from mpc_amd import engine, parse_file
c = parse_file('tests/golden/aes_128.gcf')
L = engine.lib(); batch=1024
k = np.frombuffer(bytes(range(32)), np.uint8).copy()
rnd = np.frombuffer(np.random.default_rng(1).bytes(batch * 16 * (c.num_inputs + 1)), np.uint8).copy()
for pinned in (False, True):
    P = Party(c, batch, pinned)
    for r in range(3):
        print("--- pinned", pinned, "rep", r, file=sys.stderr)
        assert L.gc_garble(P.dc.h, p(k), len(k), p(rnd), len(rnd), batch, p(P.R), None, p(P.io), p(P.slabs[0])) == 0
    P.close()
P
