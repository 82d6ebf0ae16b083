"""The RCCL entry points behind gc_comm_* are bound with dlsym (mpc_amd/csrc/comm.cpp): nothing but the function-pointer types
written in mpc_amd/csrc/rccl_binding.h says what their arguments are.  That header checks every one of them — and the sizes
and enum values that cross the C ABI — against <rccl/rccl.h> with static_asserts; this test compiles it on CPU (no GPU, no
communicator: VERDICT r4 item 4 — what a first real run with more than one rank would otherwise be the first to see)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_prototypes_match_the_header(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no hipcc / rccl.h here")
    src = tmp_path / "binding.cpp"
    src.write_text('#include "rccl_binding.h"\nint main() { gc_rccl::Rccl r; return r.ok ? 1 : 0; }\n')
    run = subprocess.run([hipcc, "-std=c++17", "-fsyntax-only", "-x", "hip", "--offload-arch=gfx950", "-I",
                          os.path.join(ROOT, "mpc_amd", "csrc"), str(src)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    # ... and the check is a real one: a prototype that differs is refused
    bad = tmp_path / "bad.cpp"
    bad.write_text('#include <rccl/rccl.h>\n#include <type_traits>\n'
                   'static_assert(std::is_same<ncclResult_t (*)(ncclUniqueId *, int), decltype(&ncclGetUniqueId)>::value, "differs");\n')
    run = subprocess.run([hipcc, "-std=c++17", "-fsyntax-only", "-x", "hip", "--offload-arch=gfx950", str(bad)],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode != 0 and "differs" in run.stderr
