"""Circuit data model + file formats (host side, mirrors package `circuit` of the reference).

  circuit.Gate / Operation / Circuit   circuit/circuit.go:22-34,120-131,260-266
  ParseBristol                         circuit/parser.go:265-494
  ParseMPCLC                           circuit/parser.go:71-211 (magic 0x63726300, marshal.go:17)

plus the repo's own compact fixture container (".gcf", see `save_gcf`) so that the GPU
box needs nothing outside /root/repo.  Pure host logic: no GPU, no oracle.
"""
import io
import re
import struct
import zlib

import numpy as np

# circuit.Operation (circuit/circuit.go:25-34)
XOR, XNOR, AND, OR, INV = range(5)
OP_NAMES = {"XOR": XOR, "XNOR": XNOR, "AND": AND, "OR": OR, "INV": INV}

# Memory layouts crossing the C ABI (== Go's in-memory structs)
LABEL = np.dtype([("d0", "<u8"), ("d1", "<u8")])  # ot.Label   ot/label.go:28-31
WIRE = np.dtype([("l0", LABEL), ("l1", LABEL)])  # ot.Wire    ot/label.go:18-21
GATE = np.dtype(  # circuit.Gate, 20 bytes (circuit_test.go:14-19)
    {
        "names": ["in0", "in1", "out", "op", "level"],
        "formats": ["<u4", "<u4", "<u4", "u1", "<u4"],
        "offsets": [0, 4, 8, 12, 16],
        "itemsize": 20,
    }
)

MPCLC_MAGIC = 0x63726300
GCF_MAGIC = b"GCF1"


class CircuitError(ValueError):
    pass


class Circuit:
    """circuit.Circuit (circuit/circuit.go:120-131): NumGates, NumWires, Inputs, Outputs, Gates."""

    def __init__(self, num_wires, inputs, outputs, gates, name=""):
        self.NumGates = int(len(gates))
        self.NumWires = int(num_wires)
        self.Inputs = [int(b) for b in inputs]  # bits per input argument (garbler first)
        self.Outputs = [int(b) for b in outputs]
        self.Gates = np.ascontiguousarray(gates, dtype=GATE)
        self.name = name
        self._engine = None  # lazily created device handle (mpc_amd.engine)

    # IO.Size()
    @property
    def num_inputs(self):
        return sum(self.Inputs)

    @property
    def num_outputs(self):
        return sum(self.Outputs)

    def stats(self):
        """circuit.Stats gate counts (circuit.go:43-98)"""
        op = self.Gates["op"]
        return {n: int(np.count_nonzero(op == v)) for n, v in OP_NAMES.items()}

    def slab_rows(self):
        """garbled-table labels per instance (garble.go:199-211): AND 2, OR 3, INV 1"""
        s = self.stats()
        return 2 * s["AND"] + 3 * s["OR"] + s["INV"]

    def compute_bits(self, input_bits):
        """Plaintext evaluation, the gate loop of (*Circuit).Compute (circuit/computer.go:42-88): input bits are the
        wires [0, Inputs.Size()) in argument order; returns the bit of every wire (outputs = the last Outputs.Size())."""
        wires = np.zeros(self.NumWires, np.uint8)
        bits = np.asarray(input_bits, np.uint8) & 1
        if len(bits) != self.num_inputs:
            raise CircuitError("invalid inputs: got %d, expected %d" % (len(bits), self.num_inputs))
        wires[: self.num_inputs] = bits
        g = self.Gates
        for i0, i1, out, op in zip(g["in0"].tolist(), g["in1"].tolist(), g["out"].tolist(), g["op"].tolist()):
            a = wires[i0]
            if op == INV:
                r = a ^ 1
            else:
                b = wires[i1]
                r = (a ^ b) if op == XOR else (a ^ b ^ 1) if op == XNOR else (a & b) if op == AND else (a | b)
            wires[out] = r
        return wires

    def __repr__(self):
        return "#gates=%d %s #w=%d" % (self.NumGates, self.stats(), self.NumWires)


def _seen_check(seen, w, what):
    if w >= len(seen):
        raise CircuitError("invalid wire %d [0...%d[" % (w, len(seen)))
    if what is not None and not seen[w]:
        raise CircuitError("input %d of gate %d not set" % (w, what))


def parse_bristol(text, name=""):
    """ParseBristol (circuit/parser.go:265-494), same validation errors."""
    if isinstance(text, bytes):
        text = text.decode()
    lines = [re.split(r"\s+", ln.strip()) for ln in text.splitlines() if ln.strip()]
    if len(lines) < 3 or len(lines[0]) != 2:
        raise CircuitError("invalid 1st line: '%s'" % (lines[0] if lines else ""))
    num_gates, num_wires = int(lines[0][0]), int(lines[0][1])
    niv = int(lines[1][0])
    if 1 + niv != len(lines[1]):
        raise CircuitError("invalid inputs line: niv=%d, len=%d" % (niv, len(lines[1])))
    inputs = [int(x) for x in lines[1][1:]]
    if sum(inputs) == 0:
        raise CircuitError("no inputs defined")
    nov = int(lines[2][0])
    if 1 + nov != len(lines[2]):
        raise CircuitError("invalid outputs line")
    outputs = [int(x) for x in lines[2][1:]]
    seen = np.zeros(num_wires, bool)
    if sum(inputs) > num_wires:
        raise CircuitError("invalid wire %d [0...%d[" % (num_wires, num_wires))
    seen[: sum(inputs)] = True
    body = lines[3:]
    if len(body) > num_gates:
        raise CircuitError("too many gates")
    gates = np.zeros(num_gates, GATE)
    for g, ln in enumerate(body):
        if len(ln) < 3:
            raise CircuitError("invalid gate: %s" % ln)
        n1, n2 = int(ln[0]), int(ln[1])
        if 2 + n1 + n2 + 1 != len(ln):
            raise CircuitError("invalid gate: %s" % ln)
        ins = [int(x) for x in ln[2 : 2 + n1]]
        outs = [int(x) for x in ln[2 + n1 : 2 + n1 + n2]]
        for w in ins:
            _seen_check(seen, w, g)
        for w in outs:
            _seen_check(seen, w, None)
            seen[w] = True
        opname = ln[-1]
        if opname not in OP_NAMES:
            raise CircuitError("invalid operation '%s'" % opname)
        op = OP_NAMES[opname]
        want = 1 if op == INV else 2
        if len(ins) != want:
            raise CircuitError("invalid number of inputs %d for %s" % (len(ins), opname))
        if len(outs) != 1:
            raise CircuitError("invalid number of outputs %d for %s" % (len(outs), opname))
        gates[g] = (ins[0], ins[1] if len(ins) > 1 else 0, outs[0], op, 0)
    if len(body) != num_gates:
        raise CircuitError("not enough gates: got %d, expected %d" % (len(body), num_gates))
    if not seen.all():
        raise CircuitError("wire %d not assigned" % int(np.argmin(seen)))
    return Circuit(num_wires, inputs, outputs, gates, name)


def _read_str(f):
    (n,) = struct.unpack(">I", f.read(4))
    return f.read(n).decode() if n else ""


def _read_ioarg(f):
    """parseIOArg (circuit/parser.go:213-245): name, type string, bits, compound[]"""
    _read_str(f)
    _read_str(f)
    bits, ncomp = struct.unpack(">II", f.read(8))
    for _ in range(ncomp):
        _read_ioarg(f)
    return bits


def parse_mpclc(data, name=""):
    """ParseMPCLC (circuit/parser.go:71-211); big-endian binary."""
    f = io.BytesIO(data)
    magic, num_gates, num_wires, nin, nout = struct.unpack(">IIIII", f.read(20))
    if magic != MPCLC_MAGIC:
        raise CircuitError("bad MPCLC magic %08x" % magic)
    inputs = [_read_ioarg(f) for _ in range(nin)]
    outputs = [_read_ioarg(f) for _ in range(nout)]
    seen = np.zeros(num_wires, bool)
    seen[: sum(inputs)] = True
    gates = np.zeros(num_gates, GATE)
    g = 0
    while True:
        b = f.read(1)
        if not b:
            break
        op = b[0]
        if g >= num_gates:
            raise CircuitError("too many gates")
        if op in (XOR, XNOR, AND, OR):
            i0, i1, o = struct.unpack(">III", f.read(12))
            _seen_check(seen, i0, g)
            _seen_check(seen, i1, g)
        elif op == INV:
            i0, o = struct.unpack(">II", f.read(8))
            i1 = 0
            _seen_check(seen, i0, g)
        else:
            raise CircuitError("unsupported gate type %d" % op)
        _seen_check(seen, o, None)
        seen[o] = True
        gates[g] = (i0, i1, o, op, 0)
        g += 1
    if g != num_gates:
        raise CircuitError("not enough gates: got %d, expected %d" % (g, num_gates))
    if not seen.all():
        raise CircuitError("wire %d not assigned" % int(np.argmin(seen)))
    return Circuit(num_wires, inputs, outputs, gates, name)


def parse_file(path):
    """circuit.Parse (circuit/parser.go:55-69): dispatch on the file suffix."""
    with open(path, "rb") as f:
        data = f.read()
    if path.endswith(".circ") or path.endswith(".bristol"):
        return parse_bristol(data, name=path)
    if path.endswith(".mpclc"):
        return parse_mpclc(data, name=path)
    if path.endswith(".gcf"):
        return load_gcf(data, name=path)
    raise CircuitError("unsupported circuit format")


# ---- .gcf: this repo's compact circuit container -------------------------------
# header: "GCF1" | u32 ngates | u32 nwires | u32 nin_args | u32 nout_args | in bits[] | out bits[]
# body (zlib): op u8[n] | in0 u32[n] | in1 u32[n] | out u32[n]   (struct-of-arrays, little endian)


def save_gcf(c):
    g = c.Gates
    head = GCF_MAGIC + struct.pack("<IIII", c.NumGates, c.NumWires, len(c.Inputs), len(c.Outputs))
    head += struct.pack("<%dI" % len(c.Inputs), *c.Inputs) + struct.pack("<%dI" % len(c.Outputs), *c.Outputs)
    body = (
        np.ascontiguousarray(g["op"]).tobytes()
        + np.ascontiguousarray(g["in0"]).tobytes()
        + np.ascontiguousarray(g["in1"]).tobytes()
        + np.ascontiguousarray(g["out"]).tobytes()
    )
    return head + zlib.compress(body, 9)


def load_gcf(data, name=""):
    if data[:4] != GCF_MAGIC:
        raise CircuitError("bad GCF magic")
    n, nw, ni, no = struct.unpack_from("<IIII", data, 4)
    off = 20
    inputs = struct.unpack_from("<%dI" % ni, data, off)
    off += 4 * ni
    outputs = struct.unpack_from("<%dI" % no, data, off)
    off += 4 * no
    body = zlib.decompress(data[off:])
    gates = np.zeros(n, GATE)
    gates["op"] = np.frombuffer(body, np.uint8, n, 0)
    gates["in0"] = np.frombuffer(body, "<u4", n, n)
    gates["in1"] = np.frombuffer(body, "<u4", n, 5 * n)
    gates["out"] = np.frombuffer(body, "<u4", n, 9 * n)
    return Circuit(nw, inputs, outputs, gates, name)


# ---- synthetic generators ---------------------------------------------------------


def and_chain(n):
    """buildANDChain (circuit/garble_bench_test.go:19-33): gate i = AND(wire i, wire i+1) -> wire i+2"""
    gates = np.zeros(n, GATE)
    gates["in0"] = np.arange(n)
    gates["in1"] = np.arange(n) + 1
    gates["out"] = np.arange(n) + 2
    gates["op"] = AND
    return Circuit(n + 2, [1, 1], [1], gates, "and_chain_%d" % n)


def synthetic_levelised(levels, width, and_frac, seed, ninputs=256, or_frac=0.0, inv_frac=0.0, xnor_frac=0.0):
    """Random levelised circuit: gate at level l draws inputs from levels < l."""
    rng = np.random.default_rng(seed)
    n = levels * width
    gates = np.zeros(n, GATE)
    lo = 0
    for l in range(levels):
        avail = ninputs + l * width
        # bias towards the previous level so that the depth really is `levels`
        prev_lo = max(0, avail - width) if l else 0
        i0 = rng.integers(prev_lo, avail, width)
        i1 = rng.integers(0, avail, width)
        u = rng.random(width)
        op = np.full(width, XOR, np.uint8)
        t = and_frac
        op[u < t] = AND
        op[(u >= t) & (u < t + or_frac)] = OR
        t += or_frac
        op[(u >= t) & (u < t + inv_frac)] = INV
        t += inv_frac
        op[(u >= t) & (u < t + xnor_frac)] = XNOR
        sl = slice(lo, lo + width)
        gates["in0"][sl] = i0
        gates["in1"][sl] = np.where(op == INV, 0, i1)
        gates["out"][sl] = ninputs + lo + np.arange(width)
        gates["op"][sl] = op
        lo += width
    nout = min(width, 128)
    return Circuit(ninputs + n, [ninputs // 2, ninputs - ninputs // 2], [nout], gates,
                   "synth_L%d_W%d_f%.2f_s%d" % (levels, width, and_frac, seed))


def _finish(gates, nw, inputs, outputs, name):
    arr = np.zeros(len(gates), GATE)
    for j, (i0, i1, o, op) in enumerate(gates):
        arr[j] = (i0, i1, o, op, 0)
    return Circuit(nw, inputs, outputs, arr, name)


def adder(bits=64):
    """a + b mod 2^bits, ripple carry with ONE AND per bit (c' = c ^ ((a ^ c) & (b ^ c)), s = a ^ b ^ c): the shape of
    the adders a compiled MPCL program streams one per SSA instruction (compiler/ssa/streamer.go:412-524).  Inputs a =
    wires [0, bits), b = [bits, 2 bits), LSB first; the sum is the LAST `bits` wires (computer.go:77-88)."""
    gates, nw = [], 2 * bits
    carry, axb = None, []
    for i in range(bits):
        a, b = i, bits + i
        x = nw; gates.append((a, b, x, XOR)); nw += 1
        axb.append((x, carry))
        if i == bits - 1:
            break
        if carry is None:
            c = nw; gates.append((a, b, c, AND)); nw += 1
        else:
            t0 = nw; gates.append((a, carry, t0, XOR)); nw += 1
            t1 = nw; gates.append((b, carry, t1, XOR)); nw += 1
            t2 = nw; gates.append((t0, t1, t2, AND)); nw += 1
            c = nw; gates.append((carry, t2, c, XOR)); nw += 1
        carry = c
    zero = None
    for x, c in axb:  # the sum bits last, in order
        if c is None:  # bit 0: s = a ^ b; re-emit through a free gate so that it is one of the last wires
            if zero is None:
                zero = nw; gates.append((0, 0, zero, XOR)); nw += 1
            o = nw; gates.append((x, zero, o, XOR)); nw += 1
        else:
            o = nw; gates.append((x, c, o, XOR)); nw += 1
    return _finish(gates, nw, [bits, bits], [bits], "adder%d" % bits)


def subtractor(bits=64):
    """a - b mod 2^bits as a + ~b + 1, ripple borrow with ONE AND per bit and the XNORs of the reference's full subtractor
    (compiler/circuits/circ_subtractor.go:17-31: w1 = XNOR(y, cin), d = XNOR(x, w1), cout = cin ^ (w1 & (x ^ cin)) — here with
    the carry of the two's-complement sum, so that bit 0 needs no constant wire: d0 = a0 ^ b0, c1 = a0 | ~b0 = ~(~a0 & b0)).
    Inputs a = wires [0, bits), b = [bits, 2 bits), LSB first; the difference is the LAST `bits` wires."""
    gates, nw = [], 2 * bits
    diffs, carry = [], None  # carry of a + ~b + 1 into bit i (None: the constant 1 into bit 0)
    for i in range(bits):
        a, b = i, bits + i
        if carry is None:
            d = nw; gates.append((a, b, d, XOR)); nw += 1          # a ^ ~b ^ 1
            if bits > 1:
                na = nw; gates.append((a, 0, na, INV)); nw += 1
                t = nw; gates.append((na, b, t, AND)); nw += 1     # borrow out of bit 0
                c = nw; gates.append((t, 0, c, INV)); nw += 1      # carry = ~borrow
        else:
            w1 = nw; gates.append((b, carry, w1, XNOR)); nw += 1    # ~b ^ c
            d = nw; gates.append((a, w1, d, XOR)); nw += 1          # a ^ ~b ^ c
            if i < bits - 1:
                w2 = nw; gates.append((a, carry, w2, XOR)); nw += 1
                w3 = nw; gates.append((w1, w2, w3, AND)); nw += 1
                c = nw; gates.append((w3, carry, c, XOR)); nw += 1
        diffs.append(d)
        if i < bits - 1:
            carry = c
    zero = nw; gates.append((0, 0, zero, XOR)); nw += 1
    for d in diffs:  # the difference bits last, in order
        o = nw; gates.append((d, zero, o, XOR)); nw += 1
    return _finish(gates, nw, [bits, bits], [bits], "subtractor%d" % bits)


def bitwise(bits, op):
    """bits independent two-input gates (op = AND / XOR / OR): r[i] = a[i] op b[i] (compiler/circuits/circ_binary.go:14-66)"""
    gates = [(i, bits + i, 2 * bits + i, op) for i in range(bits)]
    return _finish(gates, 3 * bits, [bits, bits], [bits], "bitwise%d_%d" % (bits, op))


def multiplier(bits=64):
    """a * b mod 2^bits: array multiplier (row j adds (a & b_j) << j into the running sum; only bits < `bits` are formed).
    64 bits: 2 080 partial-product ANDs + 2 016 one-AND full adders = 12.2 k gates, AND depth ~2 x bits."""
    gates, nw = [], 2 * bits
    acc = []
    for i in range(bits):  # row 0
        o = nw; gates.append((i, bits, o, AND)); nw += 1
        acc.append(o)
    for j in range(1, bits):
        carry = None
        for i in range(bits - j):  # pp bit i of row j lands on result bit i + j
            pp = nw; gates.append((i, bits + j, pp, AND)); nw += 1
            s_in = acc[i + j]
            x = nw; gates.append((s_in, pp, x, XOR)); nw += 1
            last = i == bits - j - 1
            if carry is None:
                s = x
                if not last:
                    c = nw; gates.append((s_in, pp, c, AND)); nw += 1
            else:
                s = nw; gates.append((x, carry, s, XOR)); nw += 1
                if not last:
                    t0 = nw; gates.append((s_in, carry, t0, XOR)); nw += 1
                    t1 = nw; gates.append((pp, carry, t1, XOR)); nw += 1
                    t2 = nw; gates.append((t0, t1, t2, AND)); nw += 1
                    c = nw; gates.append((carry, t2, c, XOR)); nw += 1
            acc[i + j] = s
            if not last:
                carry = c
    zero = nw; gates.append((0, 0, zero, XOR)); nw += 1
    for i in range(bits):  # the product bits last, in order
        o = nw; gates.append((acc[i], zero, o, XOR)); nw += 1
    return _finish(gates, nw, [bits, bits], [bits], "multiplier%d" % bits)


def comparator64():
    """Hand-built 64-bit unsigned a > b comparator (config 1 counterpart of millionaire.mpcl,
    apps/garbled/examples/millionaire.mpcl: `return a > b`).  Garbler input a = wires 0..63,
    evaluator input b = wires 64..127, LSB first.  gt_{i+1} = a_i & ~b_i  |  ~(a_i ^ b_i) & gt_i,
    written with XOR/AND/INV/OR so every gate type but XNOR appears."""
    gates = []
    nw = 128
    gt = None
    for i in range(64):
        a, b = i, 64 + i
        nb = nw; gates.append((b, 0, nb, INV)); nw += 1
        t = nw; gates.append((a, nb, t, AND)); nw += 1
        if gt is None:
            gt = t
            continue
        x = nw; gates.append((a, b, x, XOR)); nw += 1
        nx = nw; gates.append((x, 0, nx, INV)); nw += 1
        k = nw; gates.append((nx, gt, k, AND)); nw += 1
        o = nw; gates.append((t, k, o, OR)); nw += 1
        gt = o
    arr = np.zeros(len(gates), GATE)
    for j, (i0, i1, o, op) in enumerate(gates):
        arr[j] = (i0, i1, o, op, 0)
    return Circuit(nw, [64, 64], [1], arr, "comparator64")
