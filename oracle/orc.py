"""ctypes wrapper around oracle/liborc.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from tgsfilter_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("ORC_LIB") or os.path.join(_HERE, "liborc.so")      # (ORC_LIB: a sanitizer build, tests/manual/sanitize_emul.py)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_align_hw.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(_Aln)]
        L.orc_align_hw.restype = C.c_int
        L.orc_alignment_free.argtypes = [C.POINTER(_Aln)]
        L.orc_filter_batch.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.BatchIn), C.POINTER(abi.BatchOut),
                                       C.c_void_p, C.c_uint32]
        L.orc_filter_batch.restype = C.c_int
        _LIB = L
    return _LIB


class _Aln(C.Structure):
    _fields_ = [("ed", C.c_int), ("n", C.c_int), ("starts", C.POINTER(C.c_int)),
                ("ends", C.POINTER(C.c_int)), ("alen", C.c_int)]


def align_hw(q: bytes, t: bytes, k: int):
    """(editDistance, numLocations, starts, ends, alignmentLength) as edlibAlign(HW, PATH) reports."""
    a = _Aln()
    lib().orc_align_hw(q, len(q), t, len(t), k, C.byref(a))
    out = (a.ed, a.n, [a.starts[i] for i in range(a.n)], [a.ends[i] for i in range(a.n)], a.alen)
    lib().orc_alignment_free(C.byref(a))
    return out


def filter_batch(params: abi.Params, seq, qual, offsets, lengths=None, n_bins=None, ctr=None,
                 frag_capacity=None, qual_offsets=None):
    """Run the oracle over a CSR batch.  Returns (reads, frags, counters) numpy arrays."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(lengths) if lengths is not None else len(offsets) - 1
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        maxlen = int(lengths.max()) if n else 0
    else:
        maxlen = int(np.diff(offsets).max()) if n else 0
    if n_bins is None:
        n_bins = abi.n_bins(max(maxlen, int(params.max_read_len)))
    if ctr is None:
        ctr = np.zeros(abi.ctr_len(params.bc_len, n_bins), dtype=np.uint64)
    if frag_capacity is None:
        frag_capacity = max(16, int((lengths.sum() if lengths is not None else offsets[-1]) // 100 + n + 16))
    reads = np.zeros(n, dtype=abi.READ_RESULT_DTYPE)
    frags = np.zeros(frag_capacity, dtype=abi.FRAGMENT_DTYPE)
    if qual_offsets is not None:
        qual_offsets = np.ascontiguousarray(qual_offsets, dtype=np.uint64)
    bi = abi.BatchIn(seq.ctypes.data, qual.ctypes.data, offsets.ctypes.data,
                     lengths.ctypes.data if lengths is not None else None, n, 0, seq.size,
                     qual_offsets.ctypes.data if qual_offsets is not None else None)
    bo = abi.BatchOut(reads.ctypes.data, frags.ctypes.data, frag_capacity, 0)
    rc = lib().orc_filter_batch(C.byref(params), C.byref(bi), C.byref(bo), ctr.ctypes.data, n_bins)
    if rc != 0:
        raise RuntimeError(f"orc_filter_batch failed: {rc}")
    return reads, frags[:bo.n_frags].copy(), ctr
