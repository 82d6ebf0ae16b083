"""Host SIMD loops of the streaming evaluator's skeleton match (mpc_amd/csrc/skel_match.cpp) against plain Python, in every
form the library can pick at run time (AVX2 / SSE2 compare, byte-shuffle / plain row moves).  CPU only: the file is plain
host C++ and is compiled here on its own."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "mpc_amd", "csrc")

WRAP = r'''
#include "skel_match.h"
extern "C" int t_level() { return skel_simd::level(); }
extern "C" int t_same(const uint8_t *b, const uint8_t *r, const uint8_t *m, size_t n) { return skel_simd::same(b, r, m, n) ? 1 : 0; }
extern "C" void t_rows(const uint8_t *b, const uint32_t *off, size_t n, gc_label *dst) { skel_simd::rows(b, off, n, dst); }
'''

CHILD = r'''
import ctypes as C, sys
import numpy as np
lib = C.CDLL(sys.argv[1])
lib.t_same.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
lib.t_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
rng = np.random.default_rng(int(sys.argv[2]))
p = lambda a: a.ctypes.data_as(C.c_void_p)
checked = 0
for n in list(range(0, 70)) + [95, 96, 97, 127, 128, 129, 1000, 4099, 66304]:
    ref = rng.integers(0, 256, n + 1, dtype=np.uint8)[:n].copy()
    mask = np.where(rng.random(n) < 0.6, 0xFF, 0).astype(np.uint8)
    buf = np.where(mask == 0, rng.integers(0, 256, n + 1, dtype=np.uint8)[:n], ref).astype(np.uint8)  # differs only under mask 0
    assert lib.t_same(p(buf), p(ref), p(mask), n) == 1, ("equal block refused", n)
    live = np.flatnonzero(mask)
    for pos in ([] if not len(live) else [live[0], live[-1], live[len(live) // 2]] + list(rng.choice(live, min(5, len(live))))):
        bad = buf.copy()
        bad[pos] ^= 1 << int(rng.integers(0, 8))  # one bit of one compared byte: first, last, anywhere
        assert lib.t_same(p(bad), p(ref), p(mask), n) == 0, ("difference missed", n, int(pos))
        checked += 1
# rows: 16 bytes BE(D0) || BE(D1) at any offset -> {D0, D1} in host order
buf = rng.integers(0, 256, 5000, dtype=np.uint8)
off = np.sort(rng.choice(5000 - 16, 200, replace=False)).astype(np.uint32)
dst = np.zeros((201, 2), np.uint64)
dst[200] = 0x5A5A5A5A5A5A5A5A
lib.t_rows(p(buf), p(off), 200, p(dst))
for r in range(200):
    raw = bytes(buf[off[r]:off[r] + 16])
    assert (int(dst[r, 0]), int(dst[r, 1])) == (int.from_bytes(raw[:8], "big"), int.from_bytes(raw[8:], "big")), r
assert int(dst[200, 0]) == 0x5A5A5A5A5A5A5A5A  # nothing beyond the last row
lib.t_rows(p(buf), p(off), 0, p(dst))
print("level", lib.t_level(), "compares", checked)
'''


@pytest.fixture(scope="module")
def simd_lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("skel")
    wrap = d / "wrap.cpp"
    wrap.write_text(WRAP)
    so = d / "libskel_test.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", CSRC, str(wrap), os.path.join(CSRC, "skel_match.cpp"),
                           "-o", str(so)])
    return str(so)


@pytest.mark.parametrize("plain", [False, True])
def test_masked_compare_and_row_moves(simd_lib, plain):
    # (the form is chosen once per process: each one in a process of its own)
    env = dict(os.environ)
    env.pop("GC_STREAM_PLAIN_MATCH", None)
    if plain:
        env["GC_STREAM_PLAIN_MATCH"] = "1"
    out = subprocess.run([sys.executable, "-c", CHILD, simd_lib, "7"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    level = int(out.stdout.split()[1])
    assert (level == 0) if plain else (level in (0, 1, 2))
