import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# No torch anywhere in the suite's own process: device memory comes from the C ABI (gc_dev_alloc / gc_dev_upload /
# gc_dev_download), so libgcengine.so brings in the system ROCm runtime by itself.  (tests/test_dist_gloo.py launches
# torch.distributed.run in CHILD processes: the launcher and its gloo control plane, nothing on a GPU.)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Chain fusion of the streaming engine (mpc_amd/csrc/stream_fuse.cpp) plans a chain of steps when it meets it the second
    # time, on a thread of its own: in a product run nobody waits for a plan.  The parity tests want the fused kernels on
    # EVERY chain, deterministically: plans at first sight, on the calling thread (tests that check the default unset it).
    os.environ.setdefault("GC_STREAM_FUSE_EAGER", "1")


def pytest_sessionfinish(session, exitstatus):
    # the product path must not lean on torch: a GPU run of this suite that imported it is a failure
    if "torch" in sys.modules and not os.environ.get("GC_ALLOW_TORCH"):
        session.exitstatus = 1
        sys.stderr.write("\ntests: `torch` was imported into the test process — device memory must come from gc_dev_*\n")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def aes_circ():
    from mpc_amd.circuit import parse_file
    return parse_file(os.path.join(GOLDEN, "aes_128.gcf"))


@pytest.fixture(scope="session")
def sha_circ():
    from mpc_amd.circuit import parse_file
    return parse_file(os.path.join(GOLDEN, "sha256xor.gcf"))


@pytest.fixture(scope="session")
def add64_circ():
    from mpc_amd.circuit import parse_file
    return parse_file(os.path.join(GOLDEN, "add64.gcf"))
