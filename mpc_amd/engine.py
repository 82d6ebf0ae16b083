"""ctypes binding of the C ABI (include/gcengine.h -> mpc_amd/csrc/libgcengine.so).

This is the same boundary the Go shim binds through cgo; the tests and bench.py
drive the product exclusively through it.  There is no CPU fallback: if the HIP library is
missing or no GPU is present, the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .circuit import GATE, LABEL, WIRE

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("GC_LIB") or os.path.join(CSRC, "libgcengine.so")  # GC_LIB: developer builds
HEADER = os.path.join(os.path.dirname(HERE), "include", "gcengine.h")

ABI_VERSION = 2  # GC_ABI_VERSION of include/gcengine.h
GC_OK, GC_E_KEYSIZE, GC_E_RAND, GC_E_GATE, GC_E_ROWS, GC_E_ARG, GC_E_HIP, GC_E_NOMEM, GC_E_WIRE = (
    0, -1, -2, -3, -4, -5, -6, -7, -8)


class EngineError(RuntimeError):
    """Carries the C-ABI status; str() follows the reference's error text where one exists."""

    def __init__(self, code, what=""):
        self.code = code
        msg = lib().gc_strerror(code).decode() if _lib is not None else "status %d" % code
        detail = lib().gc_last_error().decode() if _lib is not None and code in (GC_E_HIP, GC_E_NOMEM) else ""
        super().__init__("%s%s%s" % (what + ": " if what else "", msg, " [" + detail + "]" if detail else ""))


def build(force=False):
    """Compile libgcengine.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-s", "-C", CSRC, "-j8"]
    if force:
        subprocess.check_call(cmd + ["clean"])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


class PlanInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "ngates", "nwires", "ninputs", "noutputs", "nlevels", "max_width", "slab_rows", "n_xor", "n_xnor", "n_and",
        "n_or", "n_inv", "nslots", "n_steps", "n_hash_phases", "n_fused_steps", "n_lds_slots", "n_flat_slots",
        "n_flat_outs", "n_flat_terms", "n_flat_steps", "n_flat_units")]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP extension is the product; there is no CPU fallback)" % LIB_PATH)
    # A streaming host wants 8 hardware queues (the HIP runtime's default is 4: streams that share one run one after the other;
    # mpc_amd/csrc/engine.cpp).  The HOST decides that, before its first HIP call — this binding is the host of the test-suite
    # and of bench.py; the library itself never touches the environment.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    L = C.CDLL(LIB_PATH)
    vp, u32, sz, i32 = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int
    ip = C.POINTER(C.c_int)
    sigs = {
        "gc_strerror": (C.c_char_p, [i32]),
        "gc_last_error": (C.c_char_p, []),
        "gc_abi_version": (i32, []),
        "gc_plan_create": (vp, [vp, u32, u32, u32, u32, ip]),
        "gc_plan_create_chain": (vp, [vp, vp, vp, vp, vp, vp, u32, ip]),
        "gc_plan_free": (None, [vp]),
        "gc_plan_get_info": (i32, [vp, C.POINTER(PlanInfo)]),
        "gc_plan_simulate": (i32, [vp, vp, vp]),
        "gc_plan_describe": (i32, [vp, vp, vp, vp, vp]),
        "gc_plan_fingerprint": (i32, [vp, vp]),
        "gc_device_count": (i32, []),
        "gc_ctx_create": (vp, [i32, ip]),
        "gc_ctx_destroy": (None, [vp]),
        "gc_ctx_sync": (i32, [vp]),
        "gc_ctx_stream": (vp, [vp]),
        "gc_circ_load": (vp, [vp, vp, u32, u32, u32, u32, ip]),
        "gc_circ_free": (None, [vp]),
        "gc_circ_plan": (vp, [vp]),
        "gc_circ_set_schedule": (i32, [vp, i32]),
        "gc_garble": (i32, [vp, vp, sz, vp, sz, u32, vp, vp, vp, vp]),
        "gc_eval": (i32, [vp, vp, sz, u32, vp, vp, vp, sz, vp]),
        "gc_garble_labels": (i32, [vp, vp, sz, vp, vp, vp, vp]),
        "gc_stream_create": (vp, [vp, vp, sz, vp, sz, vp, u32, ip]),
        "gc_stream_free": (None, [vp]),
        "gc_stream_get_wire": (i32, [vp, u32, vp]),
        "gc_stream_garble": (i32, [vp, vp, u32, u32, vp, u32, vp, u32, vp, sz, C.POINTER(C.c_size_t)]),
        "gc_stream_garble_begin": (i32, [vp, vp, u32, u32, vp, u32, vp, u32]),
        "gc_stream_garble_finish": (i32, [vp, vp, sz, C.POINTER(C.c_size_t)]),
        "gc_stream_garble_flush": (i32, [vp]),
        "gc_stream_intern": (i32, [vp, vp, u32, u32, u32, u32, C.POINTER(C.c_uint32)]),
        "gc_stream_garble_begin_h": (i32, [vp, u32, vp, vp]),
        "gc_stream_release": (i32, [vp, u32]),
        "gc_stream_garble_finish_view": (i32, [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
        "gc_stream_garble_finish_async": (i32, [vp, vp, sz, C.POINTER(C.c_size_t)]),
        "gc_stream_garble_copies_wait": (i32, [vp]),
        "gc_stream_stats": (i32, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "gc_stream_deep_stats": (i32, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
        "gc_stream_fuse_stats": (i32, [vp] + [C.POINTER(C.c_uint64)] * 4),
        "gc_stream_eval_fuse_stats": (i32, [vp] + [C.POINTER(C.c_uint64)] * 4),
        "gc_stream_wait_stats": (i32, [vp, C.POINTER(C.c_uint64)]),
        "gc_stream_eval_wait_stats": (i32, [vp, C.POINTER(C.c_uint64)]),
        "gc_stream_eval_dev_stats": (i32, [vp] + [C.POINTER(C.c_uint64)] * 2),
        "gc_ctx_coop_stats": (i32, [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
        "gc_ctx_pci_bus_id": (i32, [vp, C.c_char_p, sz]),
        "gc_stream_eval_deep_stats": (i32, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
        "gc_stream_eval_create": (vp, [vp, vp, sz, ip]),
        "gc_stream_eval_free": (None, [vp]),
        "gc_stream_eval_set_wire": (i32, [vp, u32, vp]),
        "gc_stream_eval_get_wire": (i32, [vp, u32, vp]),
        "gc_stream_eval_circuit": (i32, [vp, u32, u32, u32, vp, sz, C.POINTER(C.c_size_t)]),
        "gc_stream_eval_blocks": (i32, [vp, vp, sz, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
        "gc_stream_eval_stats": (i32, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "gc_batch_create": (vp, [vp, u32, ip]),
        "gc_batch_free": (None, [vp]),
        "gc_ctx_capture_begin": (i32, [vp]),
        "gc_ctx_capture_end": (i32, [vp, C.POINTER(vp)]),
        "gc_graph_launch": (i32, [vp]),
        "gc_graph_free": (None, [vp]),
        "gc_batch_stride": (u32, [vp]),
        "gc_batch_tile_instances": (u32, [vp]),
        "gc_batch_wires_in_lds": (i32, [vp]),
        "gc_batch_set_schedule": (i32, [vp, i32]),
        "gc_batch_set_graph": (i32, [vp, i32]),
        "gc_batch_set_store_all": (i32, [vp, i32]),
        "gc_batch_garble": (i32, [vp, vp, sz, vp]),
        "gc_batch_select_inputs": (i32, [vp, vp, vp]),
        "gc_batch_set_inputs": (i32, [vp, vp]),
        "gc_batch_eval": (i32, [vp, vp, sz, vp]),
        "gc_batch_decode": (i32, [vp, vp, vp, vp]),
        "gc_batch_read_r": (i32, [vp, vp]),
        "gc_batch_read_slab": (i32, [vp, vp]),
        "gc_batch_read_wires": (i32, [vp, vp]),
        "gc_batch_read_labels": (i32, [vp, vp]),
        "gc_batch_read_outputs": (i32, [vp, vp]),
        "gc_batch_write_slab": (i32, [vp, vp]),
        "gc_batch_dev_wires": (vp, [vp]),
        "gc_batch_dev_slab": (vp, [vp]),
        "gc_batch_dev_r": (vp, [vp]),
        "gc_batch_gather_outputs": (i32, [vp, vp]),
        "gc_tables_wire_bytes": (sz, [vp]),
        "gc_batch_egress_tables": (i32, [vp, vp, sz]),
        "gc_batch_ingest_tables": (i32, [vp, vp, sz, vp]),
        "gc_batch_gather_input_wires": (i32, [vp, u32, u32, vp]),
        "gc_batch_set_input_range": (i32, [vp, u32, u32, vp]),
        "gc_cot_send_pads_dev": (i32, [vp, vp, vp, vp, vp, sz, vp]),
        "gc_cot_receive_unpad_dev": (i32, [vp, vp, vp, vp, vp, sz]),
        "gc_batch_egress_tables_dense": (i32, [vp, vp, sz]),
        "gc_batch_ingest_tables_dense": (i32, [vp, vp, sz]),
        "gc_batch_last_ms": (C.c_float, [vp]),
        "gc_batch_last_launches": (u32, [vp]),
        "gc_batch_debug_profile": (i32, [vp, i32, vp]),
        "gc_iknp_receiver_create": (vp, [vp, vp, ip]),
        "gc_iknp_sender_create": (vp, [vp, vp, vp, ip]),
        "gc_iknp_free": (None, [vp]),
        "gc_iknp_u_bytes": (sz, [sz]),
        "gc_iknp_receive": (i32, [vp, vp, sz, vp, vp]),
        "gc_iknp_send": (i32, [vp, vp, sz, sz, vp]),
        "gc_iknp_receive_dev": (i32, [vp, vp, sz, vp, vp]),
        "gc_iknp_send_dev": (i32, [vp, vp, sz, vp]),
        "gc_iknp_last_ms": (C.c_float, [vp]),
        "gc_iknp_receive_bits": (i32, [vp, vp, sz, vp, vp]),
        "gc_iknp_send_bits": (i32, [vp, vp, sz, sz, vp]),
        "gc_iknp_receive_bits_dev": (i32, [vp, vp, sz, vp, vp]),
        "gc_iknp_send_bits_dev": (i32, [vp, vp, sz, vp]),
        "gc_kos_receiver_tags": (i32, [vp, vp, vp, vp, sz, vp, vp, vp, vp, vp]),
        "gc_kos_sender_check": (i32, [vp, vp, vp, sz, vp, vp, vp, vp, vp, ip]),
        "gc_kos_receiver_tags_dev": (i32, [vp, vp, vp, vp, sz, vp, vp, vp, vp, vp]),
        "gc_kos_sender_check_dev": (i32, [vp, vp, vp, sz, vp, vp, vp, vp, vp, ip]),
        "gc_mitccrh_hash": (i32, [vp, vp, C.c_uint64, vp, sz, u32]),
        "gc_cot_send_pads": (i32, [vp, vp, vp, vp, vp, sz, vp]),
        "gc_cot_receive_unpad": (i32, [vp, vp, vp, vp, vp, sz]),
        "gc_rot_send": (i32, [vp, vp, vp, vp, sz, vp]),
        "gc_rot_receive": (i32, [vp, vp, vp, sz]),
        "gc_rot_send_dev": (i32, [vp, vp, vp, vp, sz, vp]),
        "gc_rot_receive_dev": (i32, [vp, vp, vp, sz]),
        "gc_garble_wire": (i32, [vp, vp, sz, vp, sz, u32, vp, vp, vp, sz]),
        "gc_eval_wire": (i32, [vp, vp, sz, u32, vp, vp, sz, vp, vp]),
        "gc_dev_alloc": (vp, [vp, sz, ip]),
        "gc_dev_free": (None, [vp, vp]),
        "gc_dev_upload": (i32, [vp, vp, vp, sz]),
        "gc_dev_download": (i32, [vp, vp, vp, sz]),
        "gc_dev_memset": (i32, [vp, vp, i32, sz]),
        "gc_dev_copy": (i32, [vp, vp, vp, sz]),
        "gc_host_alloc": (vp, [sz]),
        "gc_host_free": (None, [vp]),
        "gc_host_register": (i32, [vp, sz]),
        "gc_host_unregister": (i32, [vp]),
        "gc_host_is_pinned": (i32, [vp]),
        "gc_comm_available": (i32, []),
        "gc_comm_version": (i32, []),
        "gc_comm_get_unique_id": (i32, [vp, sz]),
        "gc_comm_init_rank": (vp, [vp, vp, sz, i32, i32, ip]),
        "gc_comm_init_all": (i32, [vp, i32, vp]),
        "gc_comm_destroy": (None, [vp]),
        "gc_comm_rank": (i32, [vp]),
        "gc_comm_nranks": (i32, [vp]),
        "gc_comm_allgather": (i32, [vp, vp, vp, sz]),
        "gc_comm_allgather_all": (i32, [vp, i32, vp, vp, sz]),
        "gc_comm_allreduce_max": (i32, [vp, C.POINTER(C.c_double)]),
        "gc_comm_barrier": (i32, [vp]),
    }
    for name, (res, args) in sigs.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    if L.gc_abi_version() != ABI_VERSION:
        raise ImportError("libgcengine.so ABI %d != %d" % (L.gc_abi_version(), ABI_VERSION))
    return L


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _dp(d):
    """device pointer argument: a DeviceBuffer, or a raw device address (int)"""
    return C.c_void_p(d.ptr if isinstance(d, DeviceBuffer) else d)


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


def _check(rc, what):
    if rc != GC_OK:
        raise EngineError(rc, what)


def device_count():
    return lib().gc_device_count()


class PinnedArray:
    """numpy array backed by gc_host_alloc (pinned, DMA-able host memory): what the Go shim's scratch pool hands to
    gc_garble / gc_eval.  Keep the object alive as long as views of .a are in use."""

    def __init__(self, shape, dtype):
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        self.ptr = lib().gc_host_alloc(max(n, 16))
        if not self.ptr:
            raise EngineError(GC_E_NOMEM, "gc_host_alloc(%d)" % n)
        buf = (C.c_uint8 * max(n, 1)).from_address(self.ptr)
        self.a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if self.ptr:
            self.a = None
            lib().gc_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- plan (host only) ---------------------------------------------------------------------


class Plan:
    """Levelised plan of a circuit; needs no GPU (gc_plan_*)."""

    def __init__(self, gates, nwires, ninputs, noutputs):
        L = lib()
        g = np.ascontiguousarray(gates, dtype=GATE)
        st = C.c_int(0)
        self.h = L.gc_plan_create(_p(g), len(g), nwires, ninputs, noutputs, C.byref(st))
        if not self.h:
            raise EngineError(st.value, "gc_plan_create")
        self._describe(len(g))

    def _describe(self, n):
        L = lib()
        self.info = PlanInfo()
        _check(L.gc_plan_get_info(self.h, C.byref(self.info)), "gc_plan_get_info")
        self.level_of_gate = np.zeros(n, np.uint32)
        self.tweak_of_gate = np.zeros(n, np.uint32)
        self.row_of_gate = np.zeros(n + 1, np.uint32)
        self.slot_of_gate = np.zeros(n, np.uint32)
        _check(L.gc_plan_describe(self.h, _p(self.level_of_gate), _p(self.tweak_of_gate), _p(self.row_of_gate),
                                  _p(self.slot_of_gate)), "gc_plan_describe")

    @classmethod
    def chain(cls, steps):
        """the merged plan of a fused chain (gc_plan_create_chain).  steps: [(gates, nwires, nin, nout, wiring)], wiring per
        input 0xffffffff (wire store) or m << 24 | j (output j of step m); None for a step that reads the store only"""
        L = lib()
        n = len(steps)
        keep = []
        gp, wp = (C.c_void_p * n)(), (C.c_void_p * n)()
        ng, nw, ni, no = (np.zeros(n, np.uint32) for _ in range(4))
        for k, (gates, nwires, nin, nout, wiring) in enumerate(steps):
            g = np.ascontiguousarray(gates, dtype=GATE)
            w = np.full(nin, 0xFFFFFFFF, np.uint32) if wiring is None else np.ascontiguousarray(wiring, dtype=np.uint32)
            assert len(w) == nin
            keep += [g, w]
            gp[k], wp[k] = g.ctypes.data, w.ctypes.data
            ng[k], nw[k], ni[k], no[k] = len(g), nwires, nin, nout
        st = C.c_int(0)
        self = cls.__new__(cls)
        self.h = L.gc_plan_create_chain(gp, _p(ng), _p(nw), _p(ni), _p(no), wp, n, C.byref(st))
        if not self.h:
            raise EngineError(st.value, "gc_plan_create_chain")
        self._describe(int(ng.sum()))
        self.gate_base = np.concatenate([[0], np.cumsum(ng)]).astype(np.uint32)
        return self

    def fingerprint(self):
        """64-bit fingerprint of the device program of this plan (gc_plan_fingerprint), as 16 hex digits"""
        fp = C.c_uint64(0)
        _check(lib().gc_plan_fingerprint(self.h, C.byref(fp)), "gc_plan_fingerprint")
        return "%016x" % fp.value

    def simulate(self, in_bits):
        """plaintext walk of the flattened unit program (gc_plan_simulate): output bits"""
        b = np.ascontiguousarray(in_bits, dtype=np.uint8)
        assert len(b) == self.info.ninputs
        out = np.zeros(max(self.info.noutputs, 1), np.uint8)
        _check(lib().gc_plan_simulate(self.h, _p(b) if len(b) else None, _p(out)), "gc_plan_simulate")
        return out[: self.info.noutputs]

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gc_plan_free(self.h)
            self.h = None


# ---- device objects -----------------------------------------------------------------------


class Context:
    def __init__(self, device=0):
        st = C.c_int(0)
        self.h = lib().gc_ctx_create(device, C.byref(st))
        if not self.h:
            raise EngineError(st.value, "gc_ctx_create(%d)" % device)
        self.device = device

    def coop_stats(self):
        """(state of the cooperative one-instance passes: 0 unused / 1 in use / -1 off, passes that lost a workgroup and were
        done again on the device)"""
        a, b = C.c_int(0), C.c_uint64(0)
        _check(lib().gc_ctx_coop_stats(self.h, C.byref(a), C.byref(b)), "gc_ctx_coop_stats")
        return a.value, b.value

    def pci_bus_id(self):
        """the device's PCI bus id ("0000:75:00.0"): tells the ranks' GPUs apart when every process sees its own as device 0"""
        b = C.create_string_buffer(32)
        _check(lib().gc_ctx_pci_bus_id(self.h, b, 32), "gc_ctx_pci_bus_id")
        return b.value.decode()

    def sync(self):
        _check(lib().gc_ctx_sync(self.h), "gc_ctx_sync")

    @property
    def stream(self):
        return lib().gc_ctx_stream(self.h)

    def capture(self, fn):
        """record the device-resident calls fn() makes on this ctx into a Graph (gc_ctx_capture_*)"""
        _check(lib().gc_ctx_capture_begin(self.h), "gc_ctx_capture_begin")
        g = C.c_void_p()
        try:
            fn()
        except BaseException:  # leave capture mode and drop the partial graph before the error travels on
            if lib().gc_ctx_capture_end(self.h, C.byref(g)) == GC_OK and g:
                lib().gc_graph_free(g)
            raise
        _check(lib().gc_ctx_capture_end(self.h, C.byref(g)), "gc_ctx_capture_end")
        return Graph(g)

    def zeros(self, shape, dtype=np.uint8):
        """zero-filled device buffer (gc_dev_alloc + gc_dev_memset)"""
        return DeviceBuffer(self, shape, dtype, zero=True)

    def empty(self, shape, dtype=np.uint8):
        return DeviceBuffer(self, shape, dtype)

    def random_u8(self, shape, high=256, seed=0):
        """synthetic uniform bytes in [0, high) (numpy Generator on the host, uploaded once): bench / profiling inputs"""
        return DeviceBuffer(self, data=np.random.default_rng(seed).integers(0, high, shape, dtype=np.uint8))

    def to_device(self, data):
        """host bytes / array -> device buffer (gc_dev_alloc + gc_dev_upload)"""
        if isinstance(data, (bytes, bytearray, memoryview)):
            data = np.frombuffer(bytes(data), np.uint8)
        return DeviceBuffer(self, data=data)

    def close(self):
        if self.h:
            lib().gc_ctx_destroy(self.h)
            self.h = None


class DeviceBuffer:
    """gc_dev_alloc: a device buffer owned through the C ABI (no torch, no HIP binding on the host side) — what the
    device-resident calls take as d_* pointers.  .ptr is the device address (int); offsets are plain arithmetic.
    dtype / shape are host-side book-keeping for numpy() only."""

    def __init__(self, ctx, shape=None, dtype=np.uint8, data=None, zero=False):
        self.ctx = ctx
        if data is not None:
            data = np.ascontiguousarray(data)
            shape, dtype = data.shape, data.dtype
        self.dtype = np.dtype(dtype)
        self.shape = (int(shape),) if np.isscalar(shape) else tuple(int(x) for x in shape)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        st = C.c_int(0)
        self.ptr = lib().gc_dev_alloc(ctx.h, max(self.nbytes, 16), C.byref(st))
        if not self.ptr:
            raise EngineError(st.value, "gc_dev_alloc(%d)" % self.nbytes)
        if data is not None:
            self.upload(data)
        elif zero:
            self.zero()

    def upload(self, data, offset=0):
        a = np.ascontiguousarray(data)
        assert offset + a.nbytes <= self.nbytes
        _check(lib().gc_dev_upload(self.ctx.h, C.c_void_p(self.ptr + offset), _p(a) if a.nbytes else None, a.nbytes),
               "gc_dev_upload")
        return self

    def download(self, dtype=np.uint8, shape=None, offset=0, nbytes=None):
        dtype = np.dtype(dtype)
        n = (self.nbytes - offset if nbytes is None else nbytes) if shape is None else int(np.prod(shape)) * dtype.itemsize
        assert offset + n <= self.nbytes
        out = np.empty(n // dtype.itemsize, dtype)
        _check(lib().gc_dev_download(self.ctx.h, _p(out) if n else None, C.c_void_p(self.ptr + offset), n),
               "gc_dev_download")
        return out if shape is None else out.reshape(shape)

    def __add__(self, offset):
        """device address `offset` bytes into the buffer (plain pointer arithmetic)"""
        assert 0 <= offset <= self.nbytes
        return self.ptr + int(offset)

    def numpy(self):
        """the whole buffer as a host array of the buffer's dtype / shape (waits for the ctx stream)"""
        return self.download(self.dtype, self.shape)

    def zero(self, value=0):
        _check(lib().gc_dev_memset(self.ctx.h, C.c_void_p(self.ptr), value, self.nbytes), "gc_dev_memset")
        return self

    def copy_from(self, d_src, nbytes, offset=0):
        """device -> device on the ctx stream (d_src: DeviceBuffer or device address)"""
        src = d_src.ptr if isinstance(d_src, DeviceBuffer) else int(d_src)
        _check(lib().gc_dev_copy(self.ctx.h, C.c_void_p(self.ptr + offset), C.c_void_p(src), nbytes), "gc_dev_copy")

    def close(self):
        if getattr(self, "ptr", None) and self.ctx.h:
            lib().gc_dev_free(self.ctx.h, C.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Graph:
    def __init__(self, h):
        self.h = h

    def launch(self):
        _check(lib().gc_graph_launch(self.h), "gc_graph_launch")

    def close(self):
        if self.h:
            lib().gc_graph_free(self.h)
            self.h = None


class DeviceCircuit:
    """gc_circ: a circuit.Circuit uploaded to one device."""

    def __init__(self, ctx, circuit, schedule=None):
        self.ctx = ctx
        self.c = circuit
        st = C.c_int(0)
        g = np.ascontiguousarray(circuit.Gates, dtype=GATE)
        self.h = lib().gc_circ_load(ctx.h, _p(g), len(g), circuit.NumWires, circuit.num_inputs, circuit.num_outputs,
                                    C.byref(st))
        if not self.h:
            raise EngineError(st.value, "gc_circ_load")
        self.info = PlanInfo()
        _check(lib().gc_plan_get_info(lib().gc_circ_plan(self.h), C.byref(self.info)), "gc_plan_get_info")
        if schedule is not None:
            self.set_schedule(schedule)

    def set_schedule(self, schedule):
        _check(lib().gc_circ_set_schedule(self.h, schedule), "gc_circ_set_schedule")

    @property
    def tables_wire_bytes(self):
        return int(lib().gc_tables_wire_bytes(self.h))

    # -- host-buffer API: Circuit.Garble / Circuit.Eval with a batch dimension --

    def garble(self, key, rnd, batch=1, want_wires=False, want_io=True):
        """gc_garble.  Returns dict(R[batch], slab[batch,rows], wires[batch,nwires]?, io[batch,nin+nout]?)."""
        k, r = _u8(key), _u8(rnd)
        R = np.zeros(batch, LABEL)
        slab = np.zeros((batch, max(self.info.slab_rows, 1)), LABEL)
        wires = np.zeros((batch, self.c.NumWires), WIRE) if want_wires else None
        io = np.zeros((batch, self.c.num_inputs + self.c.num_outputs), WIRE) if want_io else None
        rc = lib().gc_garble(self.h, _p(k), len(k), _p(r), len(r), batch, _p(R), _p(wires), _p(io), _p(slab))
        _check(rc, "gc_garble")
        out = {"R": R, "slab": slab[:, : self.info.slab_rows]}
        if want_wires:
            out["wires"] = wires
        if want_io:
            out["io"] = io
        return out

    def eval(self, key, slab, wires=None, inputs=None, batch=1, slab_rows=None):
        """gc_eval.  wires: LABEL [batch,nwires] (in place) or inputs: LABEL [batch,ninputs].
        Returns output labels [batch,noutputs]."""
        k = _u8(key)
        slab = np.ascontiguousarray(slab, dtype=LABEL)
        rows = slab.size // batch if slab_rows is None else slab_rows
        out = np.zeros((batch, max(self.c.num_outputs, 1)), LABEL)
        if wires is not None:
            assert wires.dtype == LABEL and wires.flags.c_contiguous and wires.size == batch * self.c.NumWires
        if inputs is not None:
            inputs = np.ascontiguousarray(inputs, dtype=LABEL)
        rc = lib().gc_eval(self.h, _p(k), len(k), batch, _p(wires), _p(inputs), _p(slab) if slab.size else None,
                           rows, _p(out))
        _check(rc, "gc_eval")
        return out[:, : self.c.num_outputs]

    def garble_wire(self, key, rnd, batch=1):
        """gc_garble_wire: dict(R, io, wire[batch, tables_wire_bytes] as uint8)"""
        k, r = _u8(key), _u8(rnd)
        stride = (self.tables_wire_bytes + 3) & ~3
        R = np.zeros(batch, LABEL)
        io = np.zeros((batch, self.c.num_inputs + self.c.num_outputs), WIRE)
        wire = np.zeros((batch, stride), np.uint8)
        _check(lib().gc_garble_wire(self.h, _p(k), len(k), _p(r), len(r), batch, _p(R), _p(io), _p(wire), stride),
               "gc_garble_wire")
        return {"R": R, "io": io, "wire": wire}

    def eval_wire(self, key, wire, inputs, batch=1):
        """gc_eval_wire: output labels [batch, noutputs]; raises GC_E_ROWS on malformed headers"""
        k = _u8(key)
        w = np.ascontiguousarray(wire, dtype=np.uint8).reshape(batch, -1)
        inp = np.ascontiguousarray(inputs, dtype=LABEL)
        out = np.zeros((batch, max(self.c.num_outputs, 1)), LABEL)
        bad = C.c_uint32(0)
        _check(lib().gc_eval_wire(self.h, _p(k), len(k), batch, _p(inp), _p(w), w.shape[1], _p(out), C.byref(bad)),
               "gc_eval_wire")
        return out[:, : self.c.num_outputs]

    def close(self):
        if self.h:
            lib().gc_circ_free(self.h)
            self.h = None


class Batch:
    """gc_batch: device-resident state of `batch` instances (garbler or evaluator role)."""

    def __init__(self, dcirc, batch):
        self.dc = dcirc
        self.batch = batch
        st = C.c_int(0)
        self.h = lib().gc_batch_create(dcirc.h, batch, C.byref(st))
        if not self.h:
            raise EngineError(st.value, "gc_batch_create")
        self.stride = lib().gc_batch_stride(self.h)

    @property
    def tile_instances(self):
        return int(lib().gc_batch_tile_instances(self.h))

    @property
    def lds_wires(self):
        return bool(lib().gc_batch_wires_in_lds(self.h))

    def set_graph(self, on):
        _check(lib().gc_batch_set_graph(self.h, 1 if on else 0), "gc_batch_set_graph")

    def set_schedule(self, s):
        _check(lib().gc_batch_set_schedule(self.h, s), "gc_batch_set_schedule")

    def set_store_all(self, on):
        _check(lib().gc_batch_set_store_all(self.h, 1 if on else 0), "gc_batch_set_store_all")

    def garble(self, key, d_rnd):
        k = _u8(key)
        _check(lib().gc_batch_garble(self.h, _p(k), len(k), _dp(d_rnd)), "gc_batch_garble")

    def select_inputs(self, garbler, d_bits):
        _check(lib().gc_batch_select_inputs(self.h, garbler.h, _dp(d_bits)), "gc_batch_select_inputs")

    def set_inputs(self, d_labels):
        _check(lib().gc_batch_set_inputs(self.h, _dp(d_labels)), "gc_batch_set_inputs")

    def eval(self, key, tables):
        k = _u8(key)
        _check(lib().gc_batch_eval(self.h, _p(k), len(k), tables.h), "gc_batch_eval")

    def decode(self, evaluator, d_bits_out, d_mismatch):
        _check(lib().gc_batch_decode(self.h, evaluator.h, _dp(d_bits_out), _dp(d_mismatch)),
               "gc_batch_decode")

    def read_r(self):
        out = np.zeros(self.batch, LABEL)
        _check(lib().gc_batch_read_r(self.h, _p(out)), "gc_batch_read_r")
        return out

    def read_slab(self):
        out = np.zeros((self.batch, max(self.dc.info.slab_rows, 1)), LABEL)
        _check(lib().gc_batch_read_slab(self.h, _p(out)), "gc_batch_read_slab")
        return out[:, : self.dc.info.slab_rows]

    def read_wires(self):
        out = np.zeros((self.batch, self.dc.c.NumWires), WIRE)
        _check(lib().gc_batch_read_wires(self.h, _p(out)), "gc_batch_read_wires")
        return out

    def read_labels(self):
        out = np.zeros((self.batch, self.dc.c.NumWires), LABEL)
        _check(lib().gc_batch_read_labels(self.h, _p(out)), "gc_batch_read_labels")
        return out

    def read_outputs(self):
        out = np.zeros((self.batch, max(self.dc.c.num_outputs, 1)), LABEL)
        _check(lib().gc_batch_read_outputs(self.h, _p(out)), "gc_batch_read_outputs")
        return out[:, : self.dc.c.num_outputs]

    def write_slab(self, slab):
        s = np.ascontiguousarray(slab, dtype=LABEL)
        assert s.size == self.batch * self.dc.info.slab_rows
        _check(lib().gc_batch_write_slab(self.h, _p(s)), "gc_batch_write_slab")

    def egress_tables(self, d_out, stride):
        _check(lib().gc_batch_egress_tables(self.h, _dp(d_out), stride), "gc_batch_egress_tables")

    def gather_input_wires(self, first, count, d_out):
        _check(lib().gc_batch_gather_input_wires(self.h, first, count, _dp(d_out)), "gc_batch_gather_input_wires")

    def set_input_range(self, first, count, d_labels):
        _check(lib().gc_batch_set_input_range(self.h, first, count, _dp(d_labels)), "gc_batch_set_input_range")

    def egress_tables_dense(self, d_out, stride):
        _check(lib().gc_batch_egress_tables_dense(self.h, _dp(d_out), stride), "gc_batch_egress_tables_dense")

    def ingest_tables_dense(self, d_in, stride):
        _check(lib().gc_batch_ingest_tables_dense(self.h, _dp(d_in), stride), "gc_batch_ingest_tables_dense")

    def ingest_tables(self, d_in, stride, d_bad):
        _check(lib().gc_batch_ingest_tables(self.h, _dp(d_in), stride, _dp(d_bad)),
               "gc_batch_ingest_tables")

    def gather_outputs(self, d_out):
        _check(lib().gc_batch_gather_outputs(self.h, _dp(d_out)), "gc_batch_gather_outputs")

    def debug_profile(self, enable=True, read=False):
        out = np.zeros(16, np.uint64) if read else None
        _check(lib().gc_batch_debug_profile(self.h, 1 if enable else 0, _p(out)), "gc_batch_debug_profile")
        return out

    @property
    def last_ms(self):
        return float(lib().gc_batch_last_ms(self.h))

    @property
    def last_launches(self):
        return int(lib().gc_batch_last_launches(self.h))

    def close(self):
        if self.h:
            lib().gc_batch_free(self.h)
            self.h = None


class Stream:
    """gc_stream: NewStreaming / Streaming.Garble / GetInput (circuit/stream_garble.go)"""

    def __init__(self, ctx, key, rnd, inputs):
        k, r = _u8(key), _u8(rnd)
        inp = np.ascontiguousarray(inputs, dtype=np.uint32)
        st = C.c_int(0)
        self.h = lib().gc_stream_create(ctx.h, _p(k), len(k), _p(r), len(r), _p(inp), len(inp), C.byref(st))
        if not self.h:
            raise EngineError(st.value, "gc_stream_create")

    def get(self, w):
        out = np.zeros(1, WIRE)
        _check(lib().gc_stream_get_wire(self.h, w, _p(out)), "gc_stream_get_wire")
        return out[0]

    def garble(self, gates, nwires, in_, out_):
        g = np.ascontiguousarray(gates, dtype=GATE)
        i = np.ascontiguousarray(in_, dtype=np.uint32)
        o = np.ascontiguousarray(out_, dtype=np.uint32)
        need = len(g) * 61 + 16  # upper bound: 13 header bytes + 3 rows per gate
        buf = getattr(self, "_buf", None)
        if buf is None or len(buf) < need:  # reused across calls: a fresh 8 MB array per step costs more than the step
            buf = self._buf = np.empty(need + need // 2, np.uint8)
        n = C.c_size_t(0)
        _check(lib().gc_stream_garble(self.h, _p(g), len(g), nwires, _p(i), len(i), _p(o), len(o), _p(buf), len(buf),
                                      C.byref(n)), "gc_stream_garble")
        return buf[: n.value].tobytes()

    def intern(self, gates, nwires, nin, nout):
        """gc_stream_intern: handle of a circuit (looked up by content once)"""
        g = np.ascontiguousarray(gates, dtype=GATE)
        h = C.c_uint32(0)
        _check(lib().gc_stream_intern(self.h, _p(g), len(g), nwires, nin, nout, C.byref(h)), "gc_stream_intern")
        need = len(g) * 61 + 16  # the buffer garble_finish fills must hold the largest circuit in flight
        buf = getattr(self, "_buf", None)
        if buf is None or len(buf) < need:
            self._buf = np.empty(need + need // 2, np.uint8)
        return h.value

    def release(self, handle):
        """gc_stream_release: the interned circuit goes back to the bounded cache, the handle is invalid afterwards"""
        _check(lib().gc_stream_release(self.h, handle), "gc_stream_release")

    def garble_begin_h(self, handle, in_, out_):
        """gc_stream_garble_begin_h: queue an interned circuit"""
        i = np.ascontiguousarray(in_, dtype=np.uint32)
        o = np.ascontiguousarray(out_, dtype=np.uint32)
        _check(lib().gc_stream_garble_begin_h(self.h, handle, _p(i), _p(o)), "gc_stream_garble_begin_h")

    def flush(self):
        """gc_stream_garble_flush: launch the queued group without waiting"""
        _check(lib().gc_stream_garble_flush(self.h), "gc_stream_garble_flush")

    def stats(self):
        """(groups launched, steps that ran in groups, steps with a launch sequence of their own)"""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().gc_stream_stats(self.h, C.byref(a), C.byref(b), C.byref(c)), "gc_stream_stats")
        return a.value, b.value, c.value

    def deep_stats(self):
        """(steps that ran on a deep lane — long one-workgroup passes beside the step groups —, lanes in use)"""
        a, b = C.c_uint64(0), C.c_uint32(0)
        _check(lib().gc_stream_deep_stats(self.h, C.byref(a), C.byref(b)), "gc_stream_deep_stats")
        return a.value, b.value

    def fuse_stats(self):
        """chain fusion (gc_stream_fuse_stats): (launch units of several steps, steps in them, merged plans built, units that
        ran step by step for want of a one-workgroup plan)"""
        v = [C.c_uint64(0) for _ in range(4)]
        _check(lib().gc_stream_fuse_stats(self.h, *[C.byref(x) for x in v]), "gc_stream_fuse_stats")
        return tuple(x.value for x in v)

    def wait_stats(self):
        """units that joined the group they conflict with and wait for units of it on the device (gc_stream_wait_stats)"""
        v = C.c_uint64(0)
        _check(lib().gc_stream_wait_stats(self.h, C.byref(v)), "gc_stream_wait_stats")
        return v.value

    def garble_begin(self, gates, nwires, in_, out_):
        """gc_stream_garble_begin: queue one circuit, do not wait (up to 4 096 in flight; small independent circuits
        share a launch sequence)"""
        g = np.ascontiguousarray(gates, dtype=GATE)
        i = np.ascontiguousarray(in_, dtype=np.uint32)
        o = np.ascontiguousarray(out_, dtype=np.uint32)
        need = len(g) * 61 + 16
        buf = getattr(self, "_buf", None)
        if buf is None or len(buf) < need:
            self._buf = np.empty(need + need // 2, np.uint8)
        _check(lib().gc_stream_garble_begin(self.h, _p(g), len(g), nwires, _p(i), len(i), _p(o), len(o)),
               "gc_stream_garble_begin")

    def garble_finish(self):
        """gc_stream_garble_finish: the bytes of the oldest circuit in flight"""
        n = C.c_size_t(0)
        _check(lib().gc_stream_garble_finish(self.h, _p(self._buf), len(self._buf), C.byref(n)), "gc_stream_garble_finish")
        return self._buf[: n.value].tobytes()

    def garble_finish_async(self, dst, offset):
        """gc_stream_garble_finish_async: the bytes of the oldest circuit in flight into dst[offset:] (a numpy uint8 array the
        caller keeps alive and untouched until copies_wait) — copied by the stream's copier threads; returns the byte count"""
        n = C.c_size_t(0)
        _check(lib().gc_stream_garble_finish_async(self.h, C.c_void_p(dst.ctypes.data + offset), len(dst) - offset, C.byref(n)),
               "gc_stream_garble_finish_async")
        return n.value

    def copies_wait(self):
        _check(lib().gc_stream_garble_copies_wait(self.h), "gc_stream_garble_copies_wait")

    def garble_finish_view(self):
        """gc_stream_garble_finish_view: the same bytes without the engine's copy — read in place from its pinned staging
        (the pointer is valid until the next finish call; the copy made here is the caller's own)"""
        ptr, n = C.c_void_p(None), C.c_size_t(0)
        _check(lib().gc_stream_garble_finish_view(self.h, C.byref(ptr), C.byref(n)), "gc_stream_garble_finish_view")
        return C.string_at(ptr, n.value) if n.value else b""

    def close(self):
        if self.h:
            lib().gc_stream_free(self.h)
            self.h = None


class StreamEval:
    """gc_stream_eval: StreamEval store + the per-gate loop of one OpCircuit block (stream_evaluator.go)"""

    def __init__(self, ctx, key):
        k = _u8(key)
        st = C.c_int(0)
        self.h = lib().gc_stream_eval_create(ctx.h, _p(k), len(k), C.byref(st))
        if not self.h:
            raise EngineError(st.value, "gc_stream_eval_create")

    def set(self, w, label):
        _check(lib().gc_stream_eval_set_wire(self.h, w, _p(_lab1(label))), "gc_stream_eval_set_wire")

    def get(self, w):
        out = np.zeros(1, LABEL)
        _check(lib().gc_stream_eval_get_wire(self.h, w, _p(out)), "gc_stream_eval_get_wire")
        return (int(out[0]["d0"]), int(out[0]["d1"]))

    def circuit(self, ngates, ntmp, nwires, data):
        b = np.frombuffer(bytes(data), np.uint8) if len(data) else np.zeros(1, np.uint8)
        n = C.c_size_t(0)
        _check(lib().gc_stream_eval_circuit(self.h, ngates, ntmp, nwires, _p(b), len(data), C.byref(n)),
               "gc_stream_eval_circuit")
        return n.value

    def blocks(self, data):
        """gc_stream_eval_blocks: framed OpCircuit blocks (20-byte headers included) -> (bytes used, blocks evaluated, more)"""
        b = np.frombuffer(bytes(data), np.uint8) if len(data) else np.zeros(1, np.uint8)
        n, nb, more = C.c_size_t(0), C.c_uint32(0), C.c_int(0)
        rc = lib().gc_stream_eval_blocks(self.h, _p(b), len(data), C.byref(n), C.byref(nb), C.byref(more))
        self.last_blocks = (n.value, nb.value, bool(more.value))  # (also when a block is refused: the ones before it are done)
        _check(rc, "gc_stream_eval_blocks")
        return self.last_blocks

    def stats(self):
        """(blocks parsed gate by gate, blocks recognised by their byte skeleton)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(lib().gc_stream_eval_stats(self.h, C.byref(a), C.byref(b)), "gc_stream_eval_stats")
        return a.value, b.value

    def deep_stats(self):
        """(blocks that ran on a deep lane, lanes in use)"""
        a, b = C.c_uint64(0), C.c_uint32(0)
        _check(lib().gc_stream_eval_deep_stats(self.h, C.byref(a), C.byref(b)), "gc_stream_eval_deep_stats")
        return a.value, b.value

    def dev_stats(self):
        """(blocks the DEVICE recognised in read buffers handed to gc_stream_eval_blocks, of those: parsed by the host after all)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(lib().gc_stream_eval_dev_stats(self.h, C.byref(a), C.byref(b)), "gc_stream_eval_dev_stats")
        return a.value, b.value

    def blocks_at(self, address, nbytes):
        """gc_stream_eval_blocks on caller memory given by address (a piece of a pinned read buffer: no copy)"""
        n, nb, more = C.c_size_t(0), C.c_uint32(0), C.c_int(0)
        rc = lib().gc_stream_eval_blocks(self.h, C.c_void_p(address), nbytes, C.byref(n), C.byref(nb), C.byref(more))
        self.last_blocks = (n.value, nb.value, bool(more.value))
        _check(rc, "gc_stream_eval_blocks")
        return self.last_blocks

    def fuse_stats(self):
        """chain fusion on the evaluator's side (gc_stream_eval_fuse_stats)"""
        v = [C.c_uint64(0) for _ in range(4)]
        _check(lib().gc_stream_eval_fuse_stats(self.h, *[C.byref(x) for x in v]), "gc_stream_eval_fuse_stats")
        return tuple(x.value for x in v)

    def wait_stats(self):
        """the evaluator's counterpart of Stream.wait_stats (gc_stream_eval_wait_stats)"""
        v = C.c_uint64(0)
        _check(lib().gc_stream_eval_wait_stats(self.h, C.byref(v)), "gc_stream_eval_wait_stats")
        return v.value

    def close(self):
        if self.h:
            lib().gc_stream_eval_free(self.h)
            self.h = None


# ---- multi-GPU (gc_comm_*: RCCL all-gather of the shards' outputs) ----------------------------

COMM_ID_BYTES = 128


def comm_available():
    return bool(lib().gc_comm_available())


def comm_version():
    """ncclGetVersion of the RCCL the library opened (0: no RCCL)"""
    return int(lib().gc_comm_version())


def comm_unique_id():
    """rank 0: the 128-byte ncclUniqueId the host hands to the other ranks"""
    buf = np.zeros(COMM_ID_BYTES, np.uint8)
    _check(lib().gc_comm_get_unique_id(_p(buf), len(buf)), "gc_comm_get_unique_id")
    return buf.tobytes()


class Comm:
    """gc_comm: one rank of the output gather (one ctx = one device)"""

    def __init__(self, ctx, uid, nranks, rank, _h=None):
        self.ctx = ctx
        if _h is not None:
            self.h = _h
        else:
            u = _u8(uid)
            st = C.c_int(0)
            self.h = lib().gc_comm_init_rank(ctx.h, _p(u), len(u), nranks, rank, C.byref(st))
            if not self.h:
                raise EngineError(st.value, "gc_comm_init_rank")
        self.rank = lib().gc_comm_rank(self.h)
        self.nranks = lib().gc_comm_nranks(self.h)

    @classmethod
    def init_all(cls, ctxs):
        """one process driving len(ctxs) devices (gc_comm_init_all)"""
        n = len(ctxs)
        hs = (C.c_void_p * n)(*[c.h for c in ctxs])
        out = (C.c_void_p * n)()
        _check(lib().gc_comm_init_all(hs, n, out), "gc_comm_init_all")
        return [cls(ctxs[i], None, n, i, _h=out[i]) for i in range(n)]

    def allgather(self, d_send, d_recv, nbytes):
        _check(lib().gc_comm_allgather(self.h, _dp(d_send), _dp(d_recv), nbytes), "gc_comm_allgather")

    @staticmethod
    def allgather_all(comms, d_sends, d_recvs, nbytes):
        n = len(comms)
        hs = (C.c_void_p * n)(*[c.h for c in comms])
        ss = (C.c_void_p * n)(*[_dp(x) for x in d_sends])
        rs = (C.c_void_p * n)(*[_dp(x) for x in d_recvs])
        _check(lib().gc_comm_allgather_all(hs, n, ss, rs, nbytes), "gc_comm_allgather_all")

    def allgather_host(self, local):
        """host array [rows, ...] -> [nranks, rows, ...]; staged through gc_dev_* buffers"""
        a = np.ascontiguousarray(local)
        d_in = DeviceBuffer(self.ctx, data=a)
        d_out = DeviceBuffer(self.ctx, a.nbytes * self.nranks)
        try:
            self.allgather(d_in.ptr, d_out.ptr, a.nbytes)
            return d_out.download(a.dtype, (self.nranks,) + a.shape)
        finally:
            d_in.close()
            d_out.close()

    def allreduce_max(self, value):
        v = C.c_double(value)
        _check(lib().gc_comm_allreduce_max(self.h, C.byref(v)), "gc_comm_allreduce_max")
        return v.value

    def barrier(self):
        _check(lib().gc_comm_barrier(self.h), "gc_comm_barrier")

    def close(self):
        if self.h:
            lib().gc_comm_destroy(self.h)
            self.h = None


# ---- OT -------------------------------------------------------------------------------------


def _lab1(x):
    a = np.zeros(1, LABEL)
    a[0] = (int(x[0]), int(x[1])) if isinstance(x, (tuple, list)) else (int(x["d0"]), int(x["d1"]))
    return a


class IKNPReceiver:
    """tail of NewIKNPReceiver + receive() (ot/iknp.go:347-356, 468-511)"""

    def __init__(self, ctx, base_wires):
        bw = np.ascontiguousarray(base_wires, dtype=WIRE)
        assert len(bw) == 128
        st = C.c_int(0)
        self.h = lib().gc_iknp_receiver_create(ctx.h, _p(bw), C.byref(st))
        if not self.h:
            raise EngineError(st.value, "gc_iknp_receiver_create")

    def receive(self, b):
        bb = np.ascontiguousarray(b, dtype=np.uint8)
        n = len(bb)
        u = np.zeros(max(lib().gc_iknp_u_bytes(n), 1), np.uint8)
        res = np.zeros(max(n, 1), LABEL)
        _check(lib().gc_iknp_receive(self.h, _p(bb), n, _p(u), _p(res)), "gc_iknp_receive")
        return u[: lib().gc_iknp_u_bytes(n)].tobytes(), res[:n]

    def receive_dev(self, d_choice_packed, n, d_u_out, d_labels_out):
        """device pointers (ints); asynchronous on the ctx stream"""
        _check(lib().gc_iknp_receive_dev(self.h, _dp(d_choice_packed), n, _dp(d_u_out), _dp(d_labels_out)), "gc_iknp_receive_dev")

    @property
    def last_ms(self):
        return float(lib().gc_iknp_last_ms(self.h))

    def receive_bits_dev(self, d_choices, n, d_u_out, d_result):
        _check(lib().gc_iknp_receive_bits_dev(self.h, _dp(d_choices), n, _dp(d_u_out), _dp(d_result)), "gc_iknp_receive_bits_dev")

    def receive_bits(self, choices, n):
        ch = np.ascontiguousarray(choices, dtype=np.uint64)
        u = np.zeros(max(lib().gc_iknp_u_bytes(n), 1), np.uint8)
        res = np.zeros(max((n + 63) // 64, 1), np.uint64)
        _check(lib().gc_iknp_receive_bits(self.h, _p(ch), n, _p(u), _p(res)), "gc_iknp_receive_bits")
        return u[: lib().gc_iknp_u_bytes(n)].tobytes(), res[: (n + 63) // 64]

    def close(self):
        if self.h:
            lib().gc_iknp_free(self.h)
            self.h = None


class IKNPSender:
    """tail of NewIKNPSender + send() (ot/iknp.go:104-122, 197-226)"""

    def __init__(self, ctx, delta, k0):
        k = np.ascontiguousarray(k0, dtype=LABEL)
        assert len(k) == 128
        d = _lab1(delta)
        st = C.c_int(0)
        self.h = lib().gc_iknp_sender_create(ctx.h, _p(d), _p(k), C.byref(st))
        if not self.h:
            raise EngineError(st.value, "gc_iknp_sender_create")

    def send(self, u, n):
        ub = np.frombuffer(bytes(u), np.uint8) if len(u) else np.zeros(1, np.uint8)
        res = np.zeros(max(n, 1), LABEL)
        _check(lib().gc_iknp_send(self.h, _p(ub), len(u), n, _p(res)), "gc_iknp_send")
        return res[:n]

    def send_dev(self, d_u_in, n, d_labels_out):
        """device pointers (ints); asynchronous on the ctx stream"""
        _check(lib().gc_iknp_send_dev(self.h, _dp(d_u_in), n, _dp(d_labels_out)), "gc_iknp_send_dev")

    @property
    def last_ms(self):
        return float(lib().gc_iknp_last_ms(self.h))

    def send_bits_dev(self, d_u_in, n, d_result):
        _check(lib().gc_iknp_send_bits_dev(self.h, _dp(d_u_in), n, _dp(d_result)), "gc_iknp_send_bits_dev")

    def send_bits(self, u, n):
        ub = np.frombuffer(bytes(u), np.uint8) if len(u) else np.zeros(1, np.uint8)
        res = np.zeros(max((n + 63) // 64, 1), np.uint64)
        _check(lib().gc_iknp_send_bits(self.h, _p(ub), len(u), n, _p(res)), "gc_iknp_send_bits")
        return res[: (n + 63) // 64]

    def close(self):
        if self.h:
            lib().gc_iknp_free(self.h)
            self.h = None


def mitccrh_hash(ctx, seed, gid0, blks, h):
    b = np.ascontiguousarray(blks, dtype=LABEL).copy()
    assert len(b) % h == 0
    s = _lab1(seed)
    _check(lib().gc_mitccrh_hash(ctx.h, _p(s), gid0, _p(b), len(b) // h, h), "gc_mitccrh_hash")
    return b


def cot_send_pads(ctx, seed, delta, data, wires):
    d = np.ascontiguousarray(data, dtype=LABEL)
    w = np.ascontiguousarray(wires, dtype=WIRE)
    out = np.zeros(max(2 * len(d), 1), LABEL)
    _check(lib().gc_cot_send_pads(ctx.h, _p(_lab1(seed)), _p(_lab1(delta)), _p(d), _p(w), len(d), _p(out)),
           "gc_cot_send_pads")
    return out[: 2 * len(d)]


def cot_send_pads_dev(ctx, seed, delta, d_data, d_wires, n, d_out):
    _check(lib().gc_cot_send_pads_dev(ctx.h, _p(_lab1(seed)), _p(_lab1(delta)), _dp(d_data), _dp(d_wires),
                                      n, _dp(d_out)), "gc_cot_send_pads_dev")


def cot_receive_unpad_dev(ctx, seed, d_flags, d_sent, d_result, n):
    _check(lib().gc_cot_receive_unpad_dev(ctx.h, _p(_lab1(seed)), _dp(d_flags), _dp(d_sent),
                                          _dp(d_result), n), "gc_cot_receive_unpad_dev")


def rot_send(ctx, seed, delta, data):
    """gc_rot_send: ROT.Send's pad loop (ot/rot.go:156-172) -> WIRE[n]"""
    d = np.ascontiguousarray(data, dtype=LABEL)
    out = np.zeros(max(len(d), 1), WIRE)
    _check(lib().gc_rot_send(ctx.h, _p(_lab1(seed)), _p(_lab1(delta)), _p(d) if len(d) else None, len(d), _p(out)),
           "gc_rot_send")
    return out[: len(d)]


def rot_receive(ctx, seed, result):
    """gc_rot_receive: ROT.Receive's pad loop (ot/rot.go:194-199)"""
    r = np.ascontiguousarray(result, dtype=LABEL).copy()
    _check(lib().gc_rot_receive(ctx.h, _p(_lab1(seed)), _p(r) if len(r) else None, len(r)), "gc_rot_receive")
    return r


def rot_send_dev(ctx, seed, delta, d_data, n, d_wires_out):
    _check(lib().gc_rot_send_dev(ctx.h, _p(_lab1(seed)), _p(_lab1(delta)), _dp(d_data), n, _dp(d_wires_out)), "gc_rot_send_dev")


def rot_receive_dev(ctx, seed, d_result, n):
    _check(lib().gc_rot_receive_dev(ctx.h, _p(_lab1(seed)), _dp(d_result), n), "gc_rot_receive_dev")


def cot_receive_unpad(ctx, seed, flags, sent, result):
    f = np.ascontiguousarray(flags, dtype=np.uint8)
    s = np.ascontiguousarray(sent, dtype=LABEL)
    r = np.ascontiguousarray(result, dtype=LABEL).copy()
    _check(lib().gc_cot_receive_unpad(ctx.h, _p(_lab1(seed)), _p(f), _p(s), _p(r), len(f)), "gc_cot_receive_unpad")
    return r


def kos_receiver_tags(ctx, seed2, result, b, choice_vec, bcv):
    r = np.ascontiguousarray(result, dtype=LABEL)
    bb = np.ascontiguousarray(b, dtype=np.uint8)
    cv = np.ascontiguousarray(choice_vec, dtype=LABEL)
    bc = np.ascontiguousarray(bcv, dtype=np.uint8)
    x, t0, t1 = np.zeros(1, LABEL), np.zeros(1, LABEL), np.zeros(1, LABEL)
    _check(lib().gc_kos_receiver_tags(ctx.h, _p(_lab1(seed2)), _p(r) if len(r) else None, _p(bb) if len(r) else None,
                                      len(r), _p(cv), _p(bc), _p(x), _p(t0), _p(t1)), "gc_kos_receiver_tags")
    f = lambda a: (int(a[0]["d0"]), int(a[0]["d1"]))
    return f(x), f(t0), f(t1)


def kos_receiver_tags_dev(ctx, seed2, d_result, d_b, n, choice_vec, bcv):
    """as kos_receiver_tags with the n labels / choice bytes in HBM (device pointers)"""
    cv = np.ascontiguousarray(choice_vec, dtype=LABEL)
    bc = np.ascontiguousarray(bcv, dtype=np.uint8)
    x, t0, t1 = np.zeros(1, LABEL), np.zeros(1, LABEL), np.zeros(1, LABEL)
    _check(lib().gc_kos_receiver_tags_dev(ctx.h, _p(_lab1(seed2)), _dp(d_result), _dp(d_b), n, _p(cv), _p(bc),
                                          _p(x), _p(t0), _p(t1)), "gc_kos_receiver_tags_dev")
    f = lambda a: (int(a[0]["d0"]), int(a[0]["d1"]))
    return f(x), f(t0), f(t1)


def kos_sender_check_dev(ctx, seed2, d_result, n, choice_vec, delta, x, t0, t1):
    cv = np.ascontiguousarray(choice_vec, dtype=LABEL)
    ok = C.c_int(0)
    _check(lib().gc_kos_sender_check_dev(ctx.h, _p(_lab1(seed2)), _dp(d_result), n, _p(cv), _p(_lab1(delta)),
                                         _p(_lab1(x)), _p(_lab1(t0)), _p(_lab1(t1)), C.byref(ok)), "gc_kos_sender_check_dev")
    return bool(ok.value)


def kos_sender_check(ctx, seed2, result, choice_vec, delta, x, t0, t1):
    r = np.ascontiguousarray(result, dtype=LABEL)
    cv = np.ascontiguousarray(choice_vec, dtype=LABEL)
    ok = C.c_int(0)
    _check(lib().gc_kos_sender_check(ctx.h, _p(_lab1(seed2)), _p(r) if len(r) else None, len(r), _p(cv), _p(_lab1(delta)),
                                     _p(_lab1(x)), _p(_lab1(t0)), _p(_lab1(t1)), C.byref(ok)), "gc_kos_sender_check")
    return bool(ok.value)
