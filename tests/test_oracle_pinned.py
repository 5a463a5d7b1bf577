"""Pin the oracle (oracle/tgsf_oracle.c) against the reference.

 * edlib_vectors.json -- outputs of the reference's own edlib (HW/PATH), committed.
 * <case>.out.fq.gz / .stderr.txt -- whole-program outputs of the reference binary (-t 1).
 * if oracle/_ref/libedlib_ref.so is present (build container or shipped prebuilt), a live
   randomised comparison as well.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import orc
from tests import hostmodel
from tgsfilter_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_edlib_golden_vectors(golden_dir):
    vec = json.load(open(os.path.join(golden_dir, "edlib_vectors.json")))
    assert len(vec) >= 500
    for v in vec:
        got = orc.align_hw(v["q"].encode(), v["t"].encode(), v["k"])
        assert got == (v["ed"], v["n"], v["starts"], v["ends"], v["alen"]), v


def _live_edlib():
    path = os.path.join(ROOT, "oracle", "_ref", "libedlib_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libedlib_ref.so not built")

    class Cfg(C.Structure):
        _fields_ = [("k", C.c_int), ("mode", C.c_int), ("task", C.c_int), ("eq", C.c_void_p), ("neq", C.c_int)]

    class Res(C.Structure):
        _fields_ = [("status", C.c_int), ("editDistance", C.c_int), ("endLocations", C.POINTER(C.c_int)),
                    ("startLocations", C.POINTER(C.c_int)), ("numLocations", C.c_int),
                    ("alignment", C.POINTER(C.c_ubyte)), ("alignmentLength", C.c_int),
                    ("alphabetLength", C.c_int)]
    lib = C.CDLL(path)
    lib.edlibAlign.restype = Res
    lib.edlibAlign.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, Cfg]
    lib.edlibFreeAlignResult.argtypes = [Res]

    def run(q, t, k):
        r = lib.edlibAlign(q, len(q), t, len(t), Cfg(k, 2, 2, None, 0))
        out = (r.editDistance, r.numLocations, [r.startLocations[i] for i in range(r.numLocations)],
               [r.endLocations[i] for i in range(r.numLocations)], r.alignmentLength)
        lib.edlibFreeAlignResult(r)
        return out
    return run


def test_live_edlib_random():
    edlib = _live_edlib()
    rng = np.random.default_rng(2024)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    odd = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    n_hit = 0
    for it in range(4000):
        Q = int(rng.choice([5, 22, 28, 45, 50, 59, 64, 65, 90, 128]))
        q = (acgt if rng.random() < 0.9 else odd)[rng.integers(0, 4, Q)].tobytes()
        T = int(rng.integers(5, 30)) if it % 5 == 0 else int(rng.integers(5, 420))
        t = bytearray((acgt[rng.integers(0, 4, T)] if rng.random() < 0.8 else
                       np.frombuffer(b"AAAC", dtype=np.uint8)[rng.integers(0, 4, T)]).tobytes())
        for _ in range(int(rng.integers(0, 3))):
            m = synth.mutate(rng, q, float(rng.choice([0.0, 0.03, 0.1, 0.2, 0.3])))
            if m and rng.random() < 0.3:
                m = m[:int(rng.integers(1, len(m) + 1))]
            p = int(rng.integers(0, max(1, T)))
            t[p:p + len(m)] = m
        t = bytes(t[:max(5, min(len(t), 420))])
        k = max(0, int(rng.choice([Q - 3, Q - 34, Q - 14, 3, Q // 3, Q - 1])))
        a, b = edlib(q, t, k), orc.align_hw(q, t, k)
        n_hit += a[0] >= 0
        assert a == b, (q, t, k, a, b)
    assert n_hit > 1000


@pytest.mark.parametrize("name", hostmodel.GOLDEN_CASES)
def test_whole_program_golden(golden_dir, name):
    """oracle + host glue == the reference binary's output file and counters."""
    case = hostmodel.GoldenCase(golden_dir, name)
    p = case.params()
    seq, qual, offsets, lengths = synth.pack(case.reads)
    results, frags, ctr = orc.filter_batch(p, seq, qual, offsets, lengths)
    info = case.info
    assert len(case.reads) == info["raw_reads"]
    assert int(lengths.sum()) == info["raw_bases"]
    if name == "qc_only":
        assert case.ref_out == b""
        assert results["n_frags"].sum() == 0
        return
    out = hostmodel.format_fastq(case.reads, results, frags)
    assert out == case.ref_out
    drop = ctr[abi.CTR_DROPINFO:abi.CTR_DROPINFO + 17]
    for i, v in enumerate(info["drop"]):
        if v is not None:
            assert int(drop[i]) == v, (i, int(drop[i]), v)
    passed = frags[(frags["flags"] & abi.FF_PASS) != 0]
    assert len(passed) == info["clean_reads"]
    assert int(passed["len"].sum()) == info["clean_bases"]
