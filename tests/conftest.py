import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# torch bundles its own ROCm runtime (libamdhip64.so.7, same soname as /opt/rocm's).  Whichever copy
# is loaded first serves the whole process, so load torch's before libgcengine.so pulls in the
# system one — otherwise a later `import torch` finds "No HIP GPUs".  (bench.py does the same.)
try:  # pragma: no cover - torch is optional for the CPU suite
    import torch  # noqa: F401
except Exception:
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
