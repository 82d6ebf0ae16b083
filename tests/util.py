"""Shared helpers for the tests: deterministic byte streams and bit packing."""
import hashlib

import numpy as np


def drbg(seed, n):
    """SHA-256 counter DRBG: the byte stream injected where the reference takes an io.Reader."""
    out = bytearray()
    ctr = 0
    s = seed if isinstance(seed, bytes) else str(seed).encode()
    while len(out) < n:
        out += hashlib.sha256(s + ctr.to_bytes(8, "big")).digest()
        ctr += 1
    return bytes(out[:n])


def bits_lsb(value, nbits):
    """wire k = bit k of the integer (circuit/computer.go:36, ioarg.go:51-66)"""
    return np.array([(value >> k) & 1 for k in range(nbits)], np.uint8)


def int_from_bits(bits):
    v = 0
    for k, b in enumerate(bits):
        v |= int(b) << k
    return v


def bytes_to_bits_little(data):
    """sha2pc/bits.go:4-15"""
    return np.unpackbits(np.frombuffer(bytes(data), np.uint8), bitorder="little")


def bits_to_bytes_little(bits):
    """sha2pc/bits.go:18-29"""
    return np.packbits(np.asarray(bits, np.uint8), bitorder="little").tobytes()
