"""CPU-side checks of the product: the C-ABI library loads and exports every symbol
include/gcengine.h declares, and the host-only plan (levelisation + tweak/row prefix sums)
agrees with the oracle's serial restatement of the reference."""
import os
import re

import numpy as np
import pytest

import oracle
from mpc_amd import engine
from mpc_amd.circuit import GATE, Circuit, and_chain, comparator64, synthetic_levelised


def declared_functions():
    src = open(engine.HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = engine.lib()
    names = declared_functions()
    assert len(names) >= 45
    for n in names:
        assert hasattr(L, n), "libgcengine.so does not export %s" % n
    assert L.gc_abi_version() == engine.ABI_VERSION == 2
    assert L.gc_strerror(engine.GC_E_KEYSIZE).decode() == "crypto/aes: invalid key size"


def test_struct_sizes_match_go_layouts():
    # circuit_test.go:14-19: unsafe.Sizeof(Gate) == 20; Label 16, Wire 32
    assert engine.GATE.itemsize == 20 and engine.LABEL.itemsize == 16 and engine.WIRE.itemsize == 32


def plan_of(c):
    return engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs)


def test_plan_matches_assign_levels(aes_circ, sha_circ):
    for c, want in ((aes_circ, (308, 192)), (sha_circ, None)):
        p = plan_of(c)
        g, nl, mw = oracle.assign_levels(c.Gates, c.NumWires)
        assert (p.info.nlevels, p.info.max_width) == (nl, mw)
        if want:
            assert (nl, mw) == want
        assert (p.level_of_gate == g["level"]).all()
        assert p.info.slab_rows == c.slab_rows()


def test_plan_tweaks_and_rows_follow_the_serial_loop():
    c = synthetic_levelised(9, 33, 0.3, seed=11, ninputs=24, or_frac=0.1, inv_frac=0.15, xnor_frac=0.1)
    p = plan_of(c)
    # serial counters of garble.go:357-359,419-420,451-452 and :199-211
    idv, row = 0, 0
    for i, g in enumerate(c.Gates):
        assert p.tweak_of_gate[i] == idv and p.row_of_gate[i] == row
        op = g["op"]
        idv += {0: 0, 1: 0, 2: 2, 3: 1, 4: 1}[int(op)]
        row += {0: 0, 1: 0, 2: 2, 3: 3, 4: 1}[int(op)]
    assert p.row_of_gate[-1] == row == c.slab_rows()
    # rows also equal the oracle's gate offsets
    gr = oracle.garble(c.Gates, c.NumWires, c.num_inputs, bytes(16), bytes(16 * (c.num_inputs + 1)))
    assert (gr["gate_off"] == p.row_of_gate).all()
    # slots: a permutation of [ninputs, ninputs+ngates), level-monotone
    assert sorted(p.slot_of_gate.tolist()) == list(range(c.num_inputs, c.num_inputs + c.NumGates))
    order = np.argsort(p.slot_of_gate)
    assert (np.diff(p.level_of_gate[order].astype(np.int64)) >= 0).all()


def test_plan_rejects_bad_circuits():
    g = np.zeros(1, GATE)
    g[0] = (0, 1, 2, 9, 0)
    with pytest.raises(engine.EngineError) as e:
        engine.Plan(g, 3, 2, 1)
    assert e.value.code == engine.GC_E_GATE
    g[0] = (0, 2, 2, 0, 0)  # reads wire 2 before anything wrote it ("input %d of gate %d not set")
    with pytest.raises(engine.EngineError) as e:
        engine.Plan(g, 3, 2, 1)
    assert e.value.code == engine.GC_E_WIRE
    g[0] = (0, 1, 5, 0, 0)  # wire id out of range
    with pytest.raises(engine.EngineError) as e:
        engine.Plan(g, 3, 2, 1)
    assert e.value.code == engine.GC_E_WIRE


def test_plan_handles_wire_reuse():
    # the parsers do not enforce single assignment (parser.go:38-44): a re-written wire must read its
    # latest value in gate order.  w2 = w0^w1 ; w2 = w2 & w0 ; w3 = w2 ^ w1
    g = np.zeros(3, GATE)
    g[0] = (0, 1, 2, 0, 0)
    g[1] = (2, 0, 2, 2, 0)
    g[2] = (2, 1, 3, 0, 0)
    p = engine.Plan(g, 4, 2, 1)
    assert p.info.nlevels == 3 and p.level_of_gate.tolist() == [0, 1, 2]
    assert len(set(p.slot_of_gate.tolist())) == 3


def test_and_chain_and_comparator_plans():
    c = and_chain(50)
    p = plan_of(c)
    assert (p.info.nlevels, p.info.max_width, p.info.slab_rows) == (50, 1, 100)
    c = comparator64()
    p = plan_of(c)
    assert p.info.n_or == 63 and p.info.n_inv == 127 and p.info.n_and == 127


def test_no_gpu_means_loud_failure():
    if engine.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(engine.EngineError) as e:
        engine.Context(0)
    assert e.value.code == engine.GC_E_HIP


def _fuzz_circuit(rng, ninputs, ngates, p_xor, reuse):
    import numpy as np
    from mpc_amd.circuit import AND, GATE, INV, OR, XNOR, XOR, Circuit
    gates = np.zeros(ngates, GATE)
    nw = ninputs
    live = list(range(ninputs))
    for i in range(ngates):
        a = live[int(rng.integers(max(0, len(live) - 12), len(live)))] if rng.random() < 0.7 else live[int(rng.integers(0, len(live)))]
        b = live[int(rng.integers(0, len(live)))]
        if rng.random() < p_xor:
            op = XOR if rng.random() < 0.8 else XNOR
            if rng.random() < 0.03:
                b = a
        else:
            op = [AND, OR, INV][int(rng.integers(0, 3))]
        if nw > ninputs and rng.random() < reuse:
            out = int(rng.integers(ninputs, nw))
        else:
            out = nw
            nw += 1
            live.append(out)
        gates[i] = (a, 0 if op == INV else b, out, op, 0)
    nout = min(int(rng.integers(1, 12)), nw)
    return Circuit(nw, [ninputs // 2, ninputs - ninputs // 2], [nout], gates)


def test_flat_schedule_simulates_to_plaintext(aes_circ, sha_circ, add64_circ):
    """gc_plan_simulate walks the flattened unit program (slots, parts, stores) on plaintext bits with the kernels'
    read-before-write semantics: must equal plain evaluation — checks the planner (hash-phase order, XOR flattening,
    LDS slot recycling, lane-split items) without a GPU."""
    import numpy as np
    from mpc_amd.circuit import and_chain, comparator64, synthetic_levelised
    rng = np.random.default_rng(77)
    circs = [aes_circ, sha_circ, add64_circ, comparator64(), and_chain(50),
             synthetic_levelised(6, 300, 0.3, seed=3, ninputs=64, or_frac=0.1, inv_frac=0.1, xnor_frac=0.1),
             synthetic_levelised(40, 20, 0.05, seed=4, ninputs=40, xnor_frac=0.3),   # long XOR lists, extra XOR rounds
             synthetic_levelised(3, 2000, 0.0, seed=5, ninputs=128)]                  # no table-producing gate at all
    for k in range(40):
        circs.append(_fuzz_circuit(rng, int(rng.integers(2, 40)), int(rng.integers(1, 500)),
                                   float(rng.choice([0.0, 0.5, 0.8, 0.95, 1.0])), float(rng.choice([0.0, 0.05, 0.2]))))
    for c in circs:
        pl = engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs)
        assert pl.info.n_flat_slots != 0xffffffff
        for _ in range(3):
            b = rng.integers(0, 2, c.num_inputs).astype(np.uint8)
            want = c.compute_bits(b)[c.NumWires - c.num_outputs:]
            assert (pl.simulate(b) == want).all(), repr(c)


def test_late_schedule_when_the_early_one_does_not_fit_lds(monkeypatch):
    """a circuit whose labels all exist early and die late (an array multiplier forms its n^2 partial products from the
    inputs alone) has no LDS plan when every hashed gate runs as early as its operands allow; the planner then runs every
    hashed gate as late as its consumers allow (plan.cpp: build_flat late) — same unit program format, far fewer live
    labels — and that schedule too must walk to the plaintext result.  GC_PLAN_NO_LATE keeps the early one."""
    import numpy as np
    from mpc_amd.circuit import multiplier, synthetic_levelised
    rng = np.random.default_rng(5)
    lds_labels = (160 * 1024 - 65536 - 256) // 16  # what fits beside the AES table, stage buffers not counted
    for c in (multiplier(128), multiplier(112)):
        pl = engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs)
        assert pl.info.n_lds_slots > lds_labels          # the early (level-walking) schedule does not fit
        assert pl.info.n_flat_slots < lds_labels // 4     # the late flattened one does, with room for several instances
        for _ in range(3):
            b = rng.integers(0, 2, c.num_inputs).astype(np.uint8)
            assert (pl.simulate(b) == c.compute_bits(b)[c.NumWires - c.num_outputs:]).all()
        a, bb = int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 62))
        w = c.num_inputs // 2
        bits = np.array([(a >> i) & 1 for i in range(w)] + [(bb >> i) & 1 for i in range(w)], np.uint8)
        got = pl.simulate(bits)
        assert sum(int(v) << i for i, v in enumerate(got)) == (a * bb) % (1 << w)
    # a circuit that fits keeps the early schedule (nothing changes for the circuits every other test pins)
    c = multiplier(64)
    monkeypatch.setenv("GC_PLAN_NO_LATE", "1")
    early = engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs).info.n_flat_slots
    monkeypatch.delenv("GC_PLAN_NO_LATE")
    assert engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs).info.n_flat_slots == early
    # ... and with the switch the big multiplier keeps its early schedule (no LDS plan)
    monkeypatch.setenv("GC_PLAN_NO_LATE", "1")
    c = multiplier(128)
    assert engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs).info.n_flat_slots > lds_labels


def test_long_xor_lists_take_two_rounds(monkeypatch):
    """an accumulator chain (the anti-diagonal of an array multiplier: up to n XOR values with 2 n terms each in ONE chunk)
    used to cost a round per 16 values; block sums shared by content make it two rounds — fewer barrier-separated steps,
    the same plaintext result"""
    import numpy as np
    from mpc_amd.circuit import multiplier
    rng = np.random.default_rng(8)
    for bits, most in ((64, 190), (128, 400)):
        c = multiplier(bits)
        pl = engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs)
        assert pl.info.n_flat_steps <= most, (bits, pl.info.n_flat_steps)
        assert pl.info.n_flat_steps <= 3 * pl.info.n_hash_phases
        for _ in range(3):
            b = rng.integers(0, 2, c.num_inputs).astype(np.uint8)
            assert (pl.simulate(b) == c.compute_bits(b)[c.NumWires - c.num_outputs:]).all()
    monkeypatch.setenv("GC_PLAN_NO_TWO_LEVEL", "1")
    c = multiplier(64)
    assert engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs).info.n_flat_steps > 230


def test_xor_only_circuit_plans_in_seconds():
    """the synthetic sweep's f = 0 circuit (131 072 XOR / XNOR-free gates, 128 levels of 1 024, every value a sum of up to 256
    inputs): the two-level form of long lists used to expand each list by substituting list into list again — ten minutes of
    planning, five of the default bench run's seven; the full expansion of an XOR value is now kept once asked for.  Same
    plan (block sums shared by content, two rounds per chunk), and it still simulates to the plaintext result."""
    import time
    c = synthetic_levelised(128, 1024, 0.0, seed=104, ninputs=256)
    t0 = time.perf_counter()
    pl = engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs)
    dt = time.perf_counter() - t0
    assert dt < 120, "planning took %.0f s" % dt  # (6 s here; 680 s before)
    assert pl.info.n_hash_phases == 0 and pl.info.n_flat_slots == 21176 and pl.info.n_flat_units == 230
    rng = np.random.default_rng(9)
    for _ in range(2):
        b = rng.integers(0, 2, c.num_inputs).astype(np.uint8)
        assert (pl.simulate(b) == c.compute_bits(b)[c.NumWires - c.num_outputs:]).all()
    # a list whose raw blocks add up to a block sum it holds already: the pair cancels (x ^ x = 0) instead of staying in the list
    c = synthetic_levelised(24, 512, 0.0, seed=500, ninputs=96)
    pl = engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs)
    assert pl.info.n_flat_terms == 33964
    b = rng.integers(0, 2, c.num_inputs).astype(np.uint8)
    assert (pl.simulate(b) == c.compute_bits(b)[c.NumWires - c.num_outputs:]).all()


def test_header_is_plain_c(tmp_path):
    """include/gcengine.h is what cgo compiles: it must be valid, warning-free plain C (C99, pedantic), not just C++;
    every declared function can be referenced from C and the Go-layout structs have the documented sizes"""
    import subprocess
    names = declared_functions()
    src = tmp_path / "abi_check.c"
    body = ['#include "gcengine.h"',
            "typedef char label_is_16[(sizeof(gc_label) == 16) ? 1 : -1];",
            "typedef char wire_is_32[(sizeof(gc_wire) == 32) ? 1 : -1];",
            "typedef char gate_is_20[(sizeof(gc_gate) == 20) ? 1 : -1];",
            "typedef void (*fn)(void);", "fn table[] = {"]
    body += ["    (fn)%s," % n for n in names]
    body += ["};", "int main(void) { return sizeof table == 0; }"]
    src.write_text("\n".join(body) + "\n")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type",
                        "-I", inc, "-c", str(src), "-o", str(tmp_path / "abi_check.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_go_shim_names_only_declared_entry_points_and_covers_the_table():
    """go/ is source only (no Go toolchain here): at least every C.gc_* it calls must exist in the header with that
    spelling, and every host-facing entry point of the header must be bound by some Go stub (the device-resident
    batch API and developer aids are reached from Go only through the additive batch wrappers)"""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set(declared_functions())
    used = set()
    for path in glob.glob(os.path.join(root, "go", "*", "*.go")):
        text = open(path).read()
        assert text.startswith("//go:build gchip"), path
        assert "\n    " not in text.replace("\n    //", ""), "%s: indent with tabs (gofmt)" % path
        used |= set(re.findall(r"\bC\.(gc_[a-z0-9_]+)\(", text))
    assert used and used <= names, sorted(used - names)
    must = {"gc_ctx_create", "gc_circ_load", "gc_garble", "gc_eval", "gc_garble_wire", "gc_eval_wire", "gc_stream_create",
            "gc_stream_get_wire", "gc_stream_intern", "gc_stream_garble_begin_h", "gc_stream_garble_finish_view",
            "gc_stream_eval_create", "gc_stream_eval_set_wire",
            "gc_stream_eval_get_wire", "gc_stream_eval_circuit", "gc_iknp_receiver_create", "gc_iknp_sender_create",
            "gc_iknp_receive", "gc_iknp_send", "gc_iknp_receive_bits", "gc_iknp_send_bits", "gc_kos_receiver_tags",
            "gc_kos_sender_check", "gc_mitccrh_hash", "gc_cot_send_pads", "gc_cot_receive_unpad", "gc_host_alloc",
            "gc_comm_init_all", "gc_comm_init_rank", "gc_comm_get_unique_id", "gc_comm_allgather_all",
            "gc_comm_allgather", "gc_rot_send", "gc_rot_receive",
            # the device-resident pipeline must be reachable from Go (go/circuit/batch_hip.go): device memory through
            # the ABI and the batch calls themselves
            "gc_dev_alloc", "gc_dev_free", "gc_dev_upload", "gc_dev_download", "gc_dev_memset", "gc_batch_create",
            "gc_batch_garble", "gc_batch_select_inputs", "gc_batch_eval", "gc_batch_decode", "gc_batch_read_slab",
            "gc_batch_read_r", "gc_batch_read_outputs", "gc_ctx_capture_begin", "gc_ctx_capture_end", "gc_graph_launch"}
    assert must <= used, sorted(must - used)


def _prototypes():
    """name -> list of 'ptr' / 'scalar' per parameter, from include/gcengine.h"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "gcengine.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"\b(gc_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        name, params = m.group(1), m.group(2).strip()
        if params in ("", "void"):
            protos[name] = []
            continue
        kinds = []
        for p in params.split(","):
            kinds.append("ptr" if ("*" in p or "[" in p) else "scalar")
        protos[name] = kinds
    return protos


def _go_calls(text):
    """(name, [argument expressions]) for every C.gc_*(...) call, parentheses balanced, strings and comments skipped"""
    import re
    text = re.sub(r"//[^\n]*", "", text)
    out = []
    for m in re.finditer(r"\bC\.(gc_[a-z0-9_]+)\(", text):
        i, depth, args, cur = m.end(), 1, [], ""
        while depth:
            ch = text[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if ch == "," and depth == 1:
                args.append(cur.strip())
                cur = ""
            else:
                cur += ch
            i += 1
        if cur.strip():
            args.append(cur.strip())
        out.append((m.group(1), args))
    return out


def test_go_shim_calls_match_the_prototypes():
    """the shim has never met a compiler (no Go toolchain in the image): every C.gc_*(...) call in go/ is checked against the
    prototype in include/gcengine.h — the NUMBER of arguments, and for every argument whose kind can be read off the Go
    expression (a C.uint32_t(...) / C.size_t(...) / C.int(...) conversion or a literal is a scalar; &x, (*C.T)(...),
    unsafe.Pointer(...), nil are pointers) that it meets a parameter of that kind.  Signatures the shim keeps:
    circuit/garble.go:248, eval.go:17, stream_garble.go:41,161, ot/iknp.go:129,364."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    protos = _prototypes()
    assert len(protos) > 100 and protos["gc_garble"].count("ptr") >= 6
    scalar = re.compile(r"^(C\.(u?int(8|16|32|64)?_t|size_t|int|uint|double|float|long|ulong)\(|[0-9]|C\.GC_)")
    pointer = re.compile(r"^(&|\(\*C\.|\(\*\*C\.|unsafe\.Pointer\(|nil$|\(C\.uintptr_t\)|\(unsafe\.Pointer\))")
    bad, ncalls = [], 0
    for path in sorted(glob.glob(os.path.join(root, "go", "*", "*.go"))):
        for name, args in _go_calls(open(path).read()):
            ncalls += 1
            want = protos.get(name)
            where = "%s: C.%s" % (os.path.relpath(path, root), name)
            if want is None:
                bad.append(where + " is not declared")
                continue
            if len(args) != len(want):
                bad.append("%s takes %d arguments, the call passes %d: %s" % (where, len(want), len(args), args))
                continue
            for k, (a, w) in enumerate(zip(args, want)):
                if scalar.match(a) and w != "scalar":
                    bad.append("%s argument %d: scalar expression %r for a pointer parameter" % (where, k + 1, a))
                if pointer.match(a) and w != "ptr":
                    bad.append("%s argument %d: pointer expression %r for a scalar parameter" % (where, k + 1, a))
    assert ncalls > 100 and not bad, "\n".join(bad)


def test_every_environment_switch_of_the_library_is_documented():
    """INTEGRATION.md's table of environment switches names every getenv of libgcengine.so (and says which are test hooks)"""
    import glob
    csrc = os.path.join(os.path.dirname(engine.HEADER), "..", "mpc_amd", "csrc")
    used = set()
    for f in glob.glob(os.path.join(csrc, "*.cpp")) + glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        used |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(f).read()))
    assert len(used) >= 20
    doc = open(os.path.join(os.path.dirname(engine.HEADER), "..", "INTEGRATION.md")).read()
    table = doc[doc.index("## Environment switches of the library"):]
    missing = sorted(v for v in used if "`%s`" % v not in table)
    assert not missing, "INTEGRATION.md does not document %s" % missing
    for hook in ("GC_COOP_FORCE_TIMEOUT", "GC_STREAM_FUSE_EAGER", "GC_RCCL_PATH"):
        line = [l for l in table.splitlines() if "`%s`" % hook in l][0]
        assert "TEST HOOK" in line, hook


def test_integration_appendix_lists_every_entry_point():
    """INTEGRATION.md's appendix (scripts/abi_index.py --write) is current: it names every function the header declares — the
    same set the library exports — on the line the header declares it on"""
    import importlib.util
    root = os.path.join(os.path.dirname(engine.HEADER), "..")
    spec = importlib.util.spec_from_file_location("abi_index", os.path.join(root, "scripts", "abi_index.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    md, total = mod.markdown()
    assert sorted(mod.names()) == declared_functions() and total == len(declared_functions())
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert mod.BEGIN in doc and mod.END in doc
    have = doc[doc.index(mod.BEGIN): doc.index(mod.END) + len(mod.END)]
    assert have == md, "INTEGRATION.md's appendix is stale: run `python scripts/abi_index.py --write`"
