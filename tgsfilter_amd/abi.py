"""ctypes mirror of include/tgsf.h (structs, constants, counter-vector layout).

Pure declarations: importing this module loads no library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

ABI_VERSION = 3
MAX_ADAPTERS = 32
MAX_ADAPTER_LEN = 8192
N_DROPINFO = 17
N_QBINS = 256
BIN_WIDTH = 100
N_STAGES = 12

OK, E_INVALID, E_NO_DEVICE, E_HIP, E_CAPACITY, E_UNSUPPORTED, E_DATA = 0, -1, -2, -3, -4, -5, -6

RF_LOWQ, RF_AD5P, RF_AD3P, RF_ADMID, RF_DISCARDED = 0x01, 0x02, 0x04, 0x08, 0x10
FF_PASS, FF_REPEAT = 0x01, 0x02
MAX_ENQUEUED = 64                   # TGSF_MAX_ENQUEUED: tgsf_submit_device batches between two tgsf_wait
NFRAGS_NOT_FINAL = 0xFFFFFFFF       # TGSF_NFRAGS_NOT_FINAL: the batch is run again by tgsf_wait (candidate pool overflow)

CTR_DROPINFO, CTR_RAW_DIFFQ, CTR_CLEAN_DIFFQ, CTR_ROWS, CTR_END_TABLES = 0, 17, 273, 529, 533
(T_RAW5P_QUAL, T_RAW5P_CNT, T_RAW3P_QUAL, T_RAW3P_CNT,
 T_CLEAN5P_QUAL, T_CLEAN5P_CNT, T_CLEAN3P_QUAL, T_CLEAN3P_CNT) = range(8)
B_RAW_QUAL, B_RAW_CNT, B_CLEAN_QUAL, B_CLEAN_CNT = range(4)


class Params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("min_len", C.c_int32), ("max_len", C.c_int32),
        ("min_q", C.c_float), ("max_q", C.c_float),
        ("bc_len", C.c_int32), ("head_trim", C.c_int32), ("tail_trim", C.c_int32),
        ("end_len", C.c_int32), ("end_match_len", C.c_int32), ("mid_match_len", C.c_int32),
        ("extra_len", C.c_int32),
        ("end_sim", C.c_float), ("mid_sim", C.c_float),
        ("discard", C.c_int32), ("filter", C.c_int32), ("only_qc", C.c_int32),
        ("min_repeat", C.c_int32), ("kmer", C.c_int32), ("qtype", C.c_int32),
        ("n_adapters", C.c_int32),
        ("adapters", C.c_char_p * MAX_ADAPTERS),
        ("adapter_len", C.c_int32 * MAX_ADAPTERS),
        ("max_batch_bases", C.c_uint64),
        ("max_batch_reads", C.c_uint32),
        ("max_read_len", C.c_uint32),
        ("no_qual", C.c_int32), ("reserved", C.c_int32),
    ]


class BatchIn(C.Structure):
    _fields_ = [
        ("seq", C.c_void_p), ("qual", C.c_void_p), ("offsets", C.c_void_p), ("lengths", C.c_void_p),
        ("n_reads", C.c_uint32), ("reserved", C.c_uint32), ("n_bytes", C.c_uint64),
        ("qual_offsets", C.c_void_p),
    ]


class BatchOut(C.Structure):
    _fields_ = [
        ("reads", C.c_void_p), ("frags", C.c_void_p),
        ("frag_capacity", C.c_uint32), ("n_frags", C.c_uint32),
    ]


READ_RESULT_DTYPE = np.dtype([
    ("sum_q", "<u8"), ("flags", "<u4"), ("n_frags", "<u4"), ("frag_begin", "<u4"),
    ("trimmed", "<u4"), ("reserved0", "<i4"), ("reserved1", "<i4"),
])
FRAGMENT_DTYPE = np.dtype([
    ("sum_q", "<u8"), ("read", "<u4"), ("start", "<i4"), ("len", "<i4"), ("flags", "<u4"),
])
assert READ_RESULT_DTYPE.itemsize == 32 and FRAGMENT_DTYPE.itemsize == 24


def ctr_end_table(t: int, bc_len: int) -> int:
    return CTR_END_TABLES + t * bc_len * 5


def ctr_bin_table(b: int, bc_len: int, n_bins: int) -> int:
    return CTR_END_TABLES + 8 * bc_len * 5 + b * n_bins * 5


def ctr_len(bc_len: int, n_bins: int) -> int:
    return ctr_bin_table(4, bc_len, n_bins)


def n_bins(max_read_len: int) -> int:
    return max_read_len // BIN_WIDTH + 1


# read-type dependent defaults, src/TGSFilter.cpp:439-457
_MID_SIM = {"hifi": 0.95, "clr": 0.9, "ont": 0.9}
_END_SIM = {"hifi": 0.9, "clr": 0.8, "ont": 0.75}


def make_params(read_type: str = "ont", *, adapters=(), min_len=1000, max_len=2147483647,
                min_q=10.0, max_q=255.0, bc_len=150, head_trim=0, tail_trim=0, end_len=150,
                end_match_len=4, mid_match_len=35, extra_len=50, end_sim=None, mid_sim=None,
                discard=False, filter=True, only_qc=False, qtype=33, min_repeat=0, kmer=11,
                max_batch_bases=0, max_batch_reads=0, max_read_len=0, no_qual=False) -> Params:
    """Para_A24 defaults as CODED (src/TGSFilter.cpp:129-171, e.g. -m defaults to 4)."""
    p = Params()
    p.struct_size = C.sizeof(Params)
    p.min_len, p.max_len = max(int(min_len), 100), int(max_len)     # :232-234 clamp
    p.min_q, p.max_q = float(min_q), float(max_q)
    p.bc_len, p.head_trim, p.tail_trim = int(bc_len), int(head_trim), int(tail_trim)
    p.end_len, p.end_match_len, p.mid_match_len, p.extra_len = (
        int(end_len), int(end_match_len), int(mid_match_len), int(extra_len))
    es = _END_SIM[read_type] if end_sim is None else max(float(end_sim), 0.7)   # :315-318
    ms = _MID_SIM[read_type] if mid_sim is None else max(float(mid_sim), 0.8)   # :324-327
    p.end_sim, p.mid_sim = es, ms
    if only_qc:
        filter = False                                                          # :409-411
    p.discard, p.filter, p.only_qc = int(bool(discard)), int(bool(filter)), int(bool(only_qc))
    p.min_repeat, p.kmer, p.qtype = int(min_repeat), int(kmer), int(qtype)
    ads = [bytes(a) for a in adapters]
    if len(ads) > MAX_ADAPTERS:
        raise ValueError("too many adapters")
    p.n_adapters = len(ads)
    p._keepalive = ads
    for i, a in enumerate(ads):
        p.adapters[i] = a
        p.adapter_len[i] = len(a)
    p.max_batch_bases, p.max_batch_reads, p.max_read_len = (
        int(max_batch_bases), int(max_batch_reads), int(max_read_len))
    p.no_qual = int(bool(no_qual))
    return p
