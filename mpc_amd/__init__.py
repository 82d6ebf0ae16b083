"""mpc_amd — MI355X-native garbled-circuit engine for the markkurossi/mpc hot path.

Product code only: HIP kernels + C ABI (csrc/), the ctypes binding (engine.py) and the
host-side mirror of the reference's circuit / ot API.  Never imports oracle/.
"""
from .circuit import (AND, GATE, INV, LABEL, OR, WIRE, XNOR, XOR, Circuit, CircuitError, and_chain,  # noqa: F401
                      comparator64, load_gcf, parse_bristol, parse_file, parse_mpclc, save_gcf,
                      synthetic_levelised)
