"""C++ host mirror (include/mpc_host.hpp): compiles and links against the C ABI on CPU; the reference's
own tests re-written against it run on the GPU (tests/cpp/test_host_mirror.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
LIBDIR = os.path.join(ROOT, "mpc_amd", "csrc")


def build(tmp_path):
    exe = str(tmp_path / "test_host_mirror")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-L", LIBDIR, "-lgcengine",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_host_mirror_compiles_and_links(tmp_path):
    build(tmp_path)


@pytest.mark.gpu
def test_host_mirror_reference_tests(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-3000:] + r.stderr[-2000:]
