"""The device-resident pipeline is reachable through include/gcengine.h alone: a plain C++ host
(tests/cpp/test_device_pipeline.cpp: no torch, no HIP header) garbles + evaluates aes_128 x 1 024 with buffers from
gc_dev_alloc, checked against the oracle; and the Python binding does the same in a process that never imports torch."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_device_pipeline.cpp")
LIBDIR = os.path.join(ROOT, "mpc_amd", "csrc")
ORCDIR = os.path.join(ROOT, "oracle")


def build(tmp_path):
    exe = str(tmp_path / "test_device_pipeline")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-L", LIBDIR, "-lgcengine",
           "-L", ORCDIR, "-loracle", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + ORCDIR, "-Wl,-rpath,/opt/rocm/lib",
           "-L/opt/rocm/lib", "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def dump_gates(c, path):
    from mpc_amd.circuit import GATE
    with open(path, "wb") as f:
        f.write(np.array([c.NumGates, c.NumWires, c.num_inputs, c.num_outputs], np.uint32).tobytes())
        f.write(np.ascontiguousarray(c.Gates, dtype=GATE).tobytes())


def test_device_pipeline_host_compiles_and_links(tmp_path):
    build(tmp_path)


@pytest.mark.gpu
def test_config2_device_resident_through_the_c_abi_only(tmp_path, aes_circ):
    exe = build(tmp_path)
    gates = str(tmp_path / "aes_128.bin")
    dump_gates(aes_circ, gates)
    r = subprocess.run([exe, gates, "1024"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_ragged_batch_and_other_circuit_through_the_c_abi_only(tmp_path, add64_circ):
    exe = build(tmp_path)
    gates = str(tmp_path / "add64.bin")
    dump_gates(add64_circ, gates)
    r = subprocess.run([exe, gates, "333"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-3000:] + r.stderr[-2000:]


NO_TORCH = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import oracle
from mpc_amd import engine, parse_file
from tests.util import drbg, bits_lsb, int_from_bits
c = parse_file(os.path.join(%(root)r, "tests", "golden", "aes_128.gcf"))
key = bytes(range(32))
batch = 1024
ctx = engine.Context(0)
dc = engine.DeviceCircuit(ctx, c)
gb, ev = engine.Batch(dc, batch), engine.Batch(dc, batch)
stride = 16 * (c.num_inputs + 1)
rnd = drbg("notorch", stride * batch)
keys = [drbg("nk%%d" %% i, 16) for i in range(batch)]
pts = [drbg("np%%d" %% i, 16) for i in range(batch)]
bits = np.zeros((batch, 256), np.uint8)
for i in range(batch):
    bits[i, :128] = bits_lsb(int.from_bytes(keys[i], "big"), 128)
    bits[i, 128:] = bits_lsb(int.from_bytes(pts[i], "big"), 128)
d_rnd, d_bits = ctx.to_device(rnd), ctx.to_device(bits)
d_out, d_mis = ctx.zeros((batch, 128)), ctx.zeros(1, np.int32)
gb.garble(key, d_rnd); ev.select_inputs(gb, d_bits); ev.eval(key, gb); gb.decode(ev, d_out, d_mis)
assert int(d_mis.numpy()[0]) == 0
out = d_out.numpy()
for i in range(batch):  # size-independent property: the decoded ciphertext is AES-128(key, pt)
    assert int_from_bits(out[i]).to_bytes(16, "big") == oracle.aes_encrypt(keys[i], pts[i]), i
R, slab, outl = gb.read_r(), gb.read_slab(), ev.read_outputs()
for i in (0, 1, 63, 64, 511, 1023):
    ref = oracle.garble(c.Gates, c.NumWires, c.num_inputs, key, rnd[i * stride:(i + 1) * stride])
    assert R[i] == ref["R"] and (slab[i] == ref["slab"]).all(), i
    w = np.zeros(c.NumWires, engine.LABEL)
    w[: c.num_inputs] = np.where(bits[i].astype(bool), ref["wires"]["l1"][: c.num_inputs], ref["wires"]["l0"][: c.num_inputs])
    oracle.eval_(c.Gates, c.NumWires, key, w, ref["slab"])
    assert (outl[i] == w[c.NumWires - c.num_outputs:]).all(), i
assert "torch" not in sys.modules, "the device-resident pipeline pulled torch in"
print("ok")
'''


@pytest.mark.gpu
def test_python_binding_runs_config2_without_torch(tmp_path):
    script = tmp_path / "no_torch.py"
    script.write_text(NO_TORCH % {"root": ROOT})
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]


def test_product_callers_do_not_import_torch():
    """grep-level guard: torch may appear in bench.py / mpc_amd / scripts only inside comments and strings about the
    launcher (`python -m torch.distributed.run`), never as an import"""
    import re
    bad = []
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for d in ("mpc_amd", "scripts"):
        for fn in sorted(os.listdir(os.path.join(ROOT, d))):
            if fn.endswith(".py"):
                files.append(os.path.join(ROOT, d, fn))
    for fn in files:
        for n, line in enumerate(open(fn), 1):
            if re.match(r"\s*(import torch|from torch)", line):
                if fn.endswith(os.path.join("mpc_amd", "dist.py")) and "torch.distributed as dist" in line:
                    continue  # init_control / exchange_unique_id: the gloo control plane of the CPU tests' launches
                bad.append("%s:%d %s" % (os.path.relpath(fn, ROOT), n, line.strip()))
    assert not bad, bad
