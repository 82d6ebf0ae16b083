"""Extended differential fuzz run (not collected by pytest): 300 random circuits x schedules 1 and 2 against the oracle.
Run on a GPU box: python tests/ext_fuzz.py"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_amd import engine
from tests.test_gpu_fuzz import random_circuit, xor_tree, KEY
from tests.test_gpu_garble_eval import check_garble_eval
ctx = engine.Context(0)
bad = 0
for seed in range(300):
    rng = np.random.default_rng(5000 + seed)
    ninputs = int(rng.integers(2, 80))
    ngates = int(rng.integers(1, 3000))
    c = random_circuit(rng, ninputs, ngates, p_xor=float(rng.choice([0.0, 0.3, 0.6, 0.8, 0.9, 0.97, 1.0])),
                       reuse=float(rng.choice([0.0, 0.02, 0.1, 0.3])), nout=int(rng.integers(1, 40)))
    batch = int(rng.choice([1, 2, 5, 64, 130, 520, 1030]))
    sample = None if batch <= 130 else sorted(set(list(range(0, batch, 97)) + [batch - 1, batch - 2, 1]))
    try:
        for schedule in (1, 2):
            check_garble_eval(ctx, c, KEY, batch, "xf%d" % seed, check_all_wires=(batch <= 130), schedule=schedule, sample=sample)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, ninputs, ngates, batch, str(e)[:100])
print("done, failures:", bad)
