#!/usr/bin/env python3
"""Regenerates the circuit fixtures under tests/golden/ from the reference's data files.

Run in the build container (needs /root/reference); the GPU box only ever sees the
committed outputs.  The inputs are DATA files of the reference (Bristol / MPCLC circuit
descriptions that its own tests load); they are re-encoded into this repo's compact
struct-of-arrays ".gcf" container (mpc_amd.circuit.save_gcf), not copied.

  aes_128.gcf     <- pkg/crypto/aes/aes_128.circ      (36 663 gates, 6 400 AND)
  sha256xor.gcf   <- sha2pc/sha256xor.mpclc           (127 806 gates, 21 455 AND)
  add64.gcf       <- pkg/math/add64.circ              (small adder, ragged-level edge case)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mpc_amd.circuit import load_gcf, parse_file, save_gcf  # noqa: E402

REF = os.environ.get("MPC_REFERENCE", "/root/reference")
SOURCES = {
    "aes_128.gcf": "pkg/crypto/aes/aes_128.circ",
    "sha256xor.gcf": "sha2pc/sha256xor.mpclc",
    "add64.gcf": "pkg/math/add64.circ",
}

for out, src in SOURCES.items():
    c = parse_file(os.path.join(REF, src))
    blob = save_gcf(c)
    back = load_gcf(blob)
    assert (back.Gates == c.Gates).all() and back.NumWires == c.NumWires
    with open(os.path.join(HERE, out), "wb") as f:
        f.write(blob)
    print("%-14s %s  %d bytes" % (out, c, len(blob)))
