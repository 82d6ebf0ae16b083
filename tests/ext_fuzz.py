"""Extended differential fuzz run (not collected by pytest): N (default 300) random circuits x schedules 1 and 2 against
the oracle, batches from 1 to 16 500 instances (tiles of 1 to 64 instances per workgroup).
Run on a GPU box: python tests/ext_fuzz.py [N]"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_amd import engine
from tests.test_gpu_fuzz import random_circuit, xor_tree, KEY
from tests.test_gpu_garble_eval import check_garble_eval
ctx = engine.Context(0)
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for seed in range(N):
    rng = np.random.default_rng(5000 + seed)
    ninputs = int(rng.integers(2, 80))
    ngates = int(rng.integers(1, 3000))
    c = random_circuit(rng, ninputs, ngates, p_xor=float(rng.choice([0.0, 0.3, 0.6, 0.8, 0.9, 0.97, 1.0])),
                       reuse=float(rng.choice([0.0, 0.02, 0.1, 0.3])), nout=int(rng.integers(1, 40)))
    batch = int(rng.choice([1, 2, 5, 64, 130, 520, 1030, 2100, 4100, 16500]))
    sample = None if batch <= 130 else sorted(set(list(range(0, batch, 97)) + [batch - 1, batch - 2, 1]))
    try:
        for schedule in (1, 2):
            check_garble_eval(ctx, c, KEY, batch, "xf%d" % seed, check_all_wires=(batch <= 130), schedule=schedule, sample=sample)
    except (AssertionError, engine.EngineError) as e:
        bad += 1
        print("FAIL seed", seed, "inputs", ninputs, "gates", ngates, "batch", batch, "schedule", schedule, str(e)[:160])
print("done, failures:", bad)
