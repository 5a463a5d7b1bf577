"""ctypes binding of libtgsf.so (include/tgsf.h) -- the Python face of the C ABI.

The library is the HIP build for MI355X.  There is no CPU fallback: loading
fails loudly if the shared object is missing, and ``Context()`` raises if no HIP
device is usable.  (tests/emul builds a serial emulation of the same kernels
for logic checks on GPU-less boxes; only tests pass its path to ``load``.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libtgsf.so")
_LIBS = {}

SYMBOLS = [
    "tgsf_abi_version", "tgsf_backend", "tgsf_prepare_device", "tgsf_device_location", "tgsf_create", "tgsf_destroy", "tgsf_submit", "tgsf_submit_async", "tgsf_submit_device",
    "tgsf_wait",
    "tgsf_counters_len", "tgsf_counters", "tgsf_counters_used", "tgsf_counters_merge", "tgsf_counters_device", "tgsf_reset_counters", "tgsf_profile",
    "tgsf_stage_times", "tgsf_stage_name", "tgsf_align_windows", "tgsf_last_error",
]


class TgsfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"tgsf error {code}: {msg}")
        self.code = code


def load(path: str | None = None):
    """Loads the library: `path` if given (the tests hand the emulation's in), else $TGSF_LIB (where libtgsf.so is
    installed, if not beside this file), else the in-tree build.  A library found through the environment or the default
    must be the HIP build (tgsf_backend() begins with "hip"): a stray TGSF_LIB never turns a GPU run into a CPU run."""
    explicit = path is not None
    path = path or os.environ.get("TGSF_LIB") or DEFAULT_LIB
    if path in _LIBS:
        if not explicit and not _LIBS[path].tgsf_backend().startswith(b"hip"):
            raise RuntimeError(f"{path} is not the HIP build of libtgsf (there is no CPU fallback; is TGSF_LIB set by accident?)")
        return _LIBS[path]
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(path)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    L.tgsf_abi_version.restype = C.c_int
    L.tgsf_backend.restype = C.c_char_p
    L.tgsf_prepare_device.argtypes = [C.c_int]
    L.tgsf_device_location.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.tgsf_create.argtypes = [C.POINTER(abi.Params), C.c_int, C.POINTER(vp)]
    L.tgsf_destroy.argtypes = [vp]
    L.tgsf_destroy.restype = None
    L.tgsf_submit.argtypes = [vp, C.POINTER(abi.BatchIn), C.POINTER(abi.BatchOut)]
    L.tgsf_submit_async.argtypes = [vp, C.POINTER(abi.BatchIn), C.POINTER(abi.BatchOut)]
    L.tgsf_submit_device.argtypes = [vp, C.POINTER(abi.BatchIn), C.POINTER(abi.BatchOut), vp, vp]
    L.tgsf_wait.argtypes = [vp]
    L.tgsf_counters_len.argtypes = [vp, C.POINTER(u64), C.POINTER(i32), C.POINTER(u32)]
    L.tgsf_counters.argtypes = [vp, vp, u64]
    L.tgsf_counters_used.argtypes = [vp, vp, u64, vp]
    L.tgsf_counters_merge.argtypes = [vp, vp]
    L.tgsf_counters_device.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.tgsf_reset_counters.argtypes = [vp]
    L.tgsf_profile.argtypes = [vp, C.c_int]
    L.tgsf_stage_times.argtypes = [vp, C.POINTER(C.c_float * abi.N_STAGES), C.POINTER(u32)]
    L.tgsf_stage_name.argtypes = [C.c_int]
    L.tgsf_stage_name.restype = C.c_char_p
    L.tgsf_align_windows.argtypes = [vp, vp, u64, vp, vp, vp, vp, u32, vp, vp]
    L.tgsf_last_error.argtypes = [vp]
    L.tgsf_last_error.restype = C.c_char_p
    if L.tgsf_abi_version() != abi.ABI_VERSION:
        raise RuntimeError("libtgsf ABI version mismatch")
    if not explicit and not L.tgsf_backend().startswith(b"hip"):
        raise RuntimeError(f"{path} is the {L.tgsf_backend().decode()!r} build of libtgsf, not the HIP build "
                           "(there is no CPU fallback; is TGSF_LIB set by accident?)")
    _LIBS[path] = L
    return L


class Context:
    """One tgsf_ctx = one GPU 'worker' (replaces one TGSFilterTask worker set)."""

    def __init__(self, params: abi.Params, device: int = 0, lib_path: str | None = None):
        self.lib = load(lib_path)
        self.params = params
        h = C.c_void_p()
        rc = self.lib.tgsf_create(C.byref(params), device, C.byref(h))
        if rc != 0:
            raise TgsfError(rc, self.lib.tgsf_last_error(None).decode())
        self.h = h
        nw, bc, nb = C.c_uint64(), C.c_int32(), C.c_uint32()
        self._chk(self.lib.tgsf_counters_len(self.h, C.byref(nw), C.byref(bc), C.byref(nb)))
        self.ctr_words, self.bc_len, self.n_bins = nw.value, bc.value, nb.value

    def _chk(self, rc):
        if rc != 0:
            raise TgsfError(rc, self.lib.tgsf_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.tgsf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer path -------------------------------------------------
    def submit(self, seq, qual, offsets, lengths=None, frag_capacity=None, qual_offsets=None):
        """Filter one CSR batch held in host memory; returns (reads, frags) structured arrays."""
        self.submit_async(seq, qual, offsets, lengths, frag_capacity, qual_offsets)
        return self.wait_result()

    def submit_async(self, seq, qual, offsets, lengths=None, frag_capacity=None, qual_offsets=None):
        """Enqueue one host batch (tgsf_submit_async); wait_result() completes it and returns the arrays."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if lengths is not None:
            lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
            n = len(lengths)
            total = int(lengths.sum())
        else:
            n = len(offsets) - 1
            total = int(offsets[-1] - offsets[0])
        if frag_capacity is None:
            frag_capacity = total // 100 + n + 16
        reads = np.zeros(n, dtype=abi.READ_RESULT_DTYPE)
        frags = np.zeros(frag_capacity, dtype=abi.FRAGMENT_DTYPE)
        if qual_offsets is not None:
            qual_offsets = np.ascontiguousarray(qual_offsets, dtype=np.uint64)
        same = qual is seq or (qual.ctypes.data == seq.ctypes.data)
        bi = abi.BatchIn(seq.ctypes.data, seq.ctypes.data if same else qual.ctypes.data, offsets.ctypes.data,
                         lengths.ctypes.data if lengths is not None else None, n, 0, seq.size,
                         qual_offsets.ctypes.data if qual_offsets is not None else None)
        bo = abi.BatchOut(reads.ctypes.data, frags.ctypes.data, frag_capacity, 0)
        # everything the library may still read or write stays referenced until wait_result()
        self._pending = (reads, frags, bi, bo, seq, qual, offsets, lengths, qual_offsets)
        rc = self.lib.tgsf_submit_async(self.h, C.byref(bi), C.byref(bo))
        if rc != 0:
            self._pending = None
            self._chk(rc)

    def wait_result(self):
        pend, self._pending = self._pending, None
        if pend is None:
            raise RuntimeError("no batch pending")
        self._chk(self.lib.tgsf_wait(self.h))
        reads, frags, _, bo = pend[:4]
        return reads, frags[:bo.n_frags].copy()

    # ---- device-resident path (pointers already in HBM) --------------------
    def submit_device(self, d_seq, d_qual, d_offsets, d_lengths, n_reads, n_bytes, d_reads, d_frags,
                      frag_capacity, d_nfrags=None, stream=None, d_qual_offsets=None):
        bi = abi.BatchIn(d_seq, d_qual, d_offsets, d_lengths, n_reads, 0, n_bytes, d_qual_offsets)
        bo = abi.BatchOut(d_reads, d_frags, frag_capacity, 0)
        self._chk(self.lib.tgsf_submit_device(self.h, C.byref(bi), C.byref(bo), d_nfrags, stream))

    def wait(self):
        self._chk(self.lib.tgsf_wait(self.h))

    def counters(self) -> np.ndarray:
        out = np.zeros(self.ctr_words, dtype=np.uint64)
        self._chk(self.lib.tgsf_counters(self.h, out.ctypes.data, self.ctr_words))
        return out

    def counters_used(self):
        """Like counters(), but only the rows in use of the bin tables travel (tgsf_counters_used)."""
        out = np.zeros(self.ctr_words, dtype=np.uint64)
        rows = (C.c_uint64 * 2)()
        self._chk(self.lib.tgsf_counters_used(self.h, out.ctypes.data, self.ctr_words, rows))
        return out, (rows[0], rows[1])

    def merge_from(self, other: "Context"):
        """Add the tallies of another context of this device to this one's, in HBM (tgsf_counters_merge)."""
        self._chk(self.lib.tgsf_counters_merge(self.h, other.h))

    def counters_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.tgsf_counters_device(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def reset_counters(self):
        self._chk(self.lib.tgsf_reset_counters(self.h))

    def profile(self, enable=True):
        self._chk(self.lib.tgsf_profile(self.h, int(enable)))

    def stage_times(self):
        arr = (C.c_float * abi.N_STAGES)()
        nb = C.c_uint32()
        self._chk(self.lib.tgsf_stage_times(self.h, C.byref(arr), C.byref(nb)))
        names = [self.lib.tgsf_stage_name(i).decode() for i in range(abi.N_STAGES)]
        return dict(zip(names, list(arr))), nb.value

    def align_windows(self, seq: bytes, win_off, win_len, adapter_id, k):
        """edlib-compatible alignments (HW/PATH) of context adapters against windows of ``seq``."""
        seq = np.frombuffer(seq, dtype=np.uint8) if isinstance(seq, (bytes, bytearray)) else np.ascontiguousarray(seq, np.uint8)
        win_off = np.ascontiguousarray(win_off, dtype=np.uint64)
        win_len = np.ascontiguousarray(win_len, dtype=np.uint32)
        adapter_id = np.ascontiguousarray(adapter_id, dtype=np.uint8)
        k = np.ascontiguousarray(k, dtype=np.int32)
        n = len(win_off)
        res = np.zeros((n, 4), dtype=np.int32)
        ends = np.zeros((n, 2), dtype=np.int32)
        self._chk(self.lib.tgsf_align_windows(self.h, seq.ctypes.data, seq.size, win_off.ctypes.data,
                                              win_len.ctypes.data, adapter_id.ctypes.data, k.ctypes.data, n,
                                              res.ctypes.data, ends.ctypes.data))
        return res, ends
