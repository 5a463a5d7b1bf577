"""ctypes binding of libtgsf_rccl.so (include/tgsf_rccl.h) and of the three RCCL calls a launcher needs.

One process per GPU: rank 0 draws an ncclUniqueId, the launcher hands it to the other ranks (bench.py broadcasts it
through its host-side gloo group), every rank calls comm_init_rank, and after its last batch
allreduce_counters(ctx, comm, rank, world) -- the job's one collective (replaces the per-thread merge of
src/TGSFilter.cpp:3208-3213 / :2673-2725 / :2586-2597 across GPUs)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
UNIQUE_ID_BYTES = 128          # ncclUniqueId: char internal[128]
_lib = None
_rccl = None


class _UniqueId(C.Structure):              # typedef struct { char internal[128]; } ncclUniqueId -- passed BY VALUE
    _fields_ = [("internal", C.c_char * UNIQUE_ID_BYTES)]


class RcclError(RuntimeError):
    pass


def load():
    global _lib, _rccl
    if _lib is None:
        _rccl = C.CDLL("librccl.so", mode=C.RTLD_GLOBAL)
        _rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        _rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        _rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        _rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _rccl.ncclGetErrorString.restype = C.c_char_p
        _lib = C.CDLL(os.path.join(_HERE, "libtgsf_rccl.so"))
        _lib.tgsf_rccl_allreduce_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.tgsf_rccl_last_error.restype = C.c_char_p
    return _lib


def unique_id() -> bytes:
    load()
    uid = _UniqueId()
    rc = _rccl.ncclGetUniqueId(C.byref(uid))
    if rc != 0:
        raise RcclError("ncclGetUniqueId: " + _rccl.ncclGetErrorString(rc).decode())
    return bytes(bytearray(uid))


def comm_init_rank(uid: bytes, rank: int, world: int):
    """The calling process must have selected its GPU (hipSetDevice / torch.cuda.set_device) beforehand."""
    load()
    comm = C.c_void_p()
    idv = _UniqueId.from_buffer_copy(uid)
    rc = _rccl.ncclCommInitRank(C.byref(comm), world, idv, rank)
    if rc != 0:
        raise RcclError("ncclCommInitRank: " + _rccl.ncclGetErrorString(rc).decode())
    return comm


def comm_count(comm) -> int:
    """Ranks RCCL itself sees in the communicator (ncclCommCount): evidence that the job's collective spans the job."""
    load()
    n = C.c_int(0)
    rc = _rccl.ncclCommCount(comm, C.byref(n))
    if rc != 0:
        raise RcclError("ncclCommCount: " + _rccl.ncclGetErrorString(rc).decode())
    return int(n.value)


def comm_destroy(comm):
    if comm:
        _rccl.ncclCommDestroy(comm)


def allreduce_counters(ctx, comm, rank: int, world: int, check_layout: bool = True, stream: int = 0):
    """ctx: tgsfilter_amd.capi.Context.  Afterwards ctx.counters() returns the totals of the whole job."""
    lib = load()
    rc = lib.tgsf_rccl_allreduce_counters(ctx.h, comm, rank, world, 1 if check_layout else 0, C.c_void_p(stream))
    if rc != 0:
        raise RcclError("tgsf_rccl_allreduce_counters (%d): %s" % (rc, lib.tgsf_rccl_last_error().decode()))
