#!/usr/bin/env python3
"""Wall time with an unaligned BAM input against the reference, same box: tests/manual/e2e_bam_check.py [n_reads]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import bamio  # noqa: E402
from tgsfilter_amd import synth  # noqa: E402
import struct, zlib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(5)
codes = np.array([1, 2, 4, 8], dtype=np.uint8)
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    bam = os.path.join(td, "in.bam")
    body = bytearray(b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", 0))
    bases = 0
    for i in range(n):
        L = int(np.clip(rng.normal(18000, 3000), 1000, 40000)) & ~1
        c = codes[rng.integers(0, 4, L)]
        packed = ((c[0::2] << 4) | c[1::2]).astype(np.uint8).tobytes()
        q = np.clip(np.rint(rng.normal(30, 6, L)), 2, 60).astype(np.uint8).tobytes()
        rn = b"read%d\0" % i
        core = struct.pack("<iiBBHHHIiii", -1, -1, len(rn), 255, 4680, 0, 4, L, -1, -1, 0)
        rec = core + rn + packed + q
        body += struct.pack("<i", len(rec)) + rec
        bases += L
    open(bam, "wb").write(bamio.bgzf(bytes(body)))
    print("bam %.0f MB, %.1f Mbases" % (os.path.getsize(bam) / 1e6, bases / 1e6))
    fa = os.path.join(td, "ad.fa"); open(fa, "wb").write(b">blunt\n" + synth.PACBIO_BLUNT + b"\n")
    flags = ["-x", "hifi", "-l", "1000", "-q", "20", "-5", "0", "-3", "0", "-a", fa]
    cores = max(1, min((os.cpu_count() or 2) - 1, 32))
    outs = {}
    for name, exe in (("reference", os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")), ("mi355x", os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter"))):
        out = os.path.join(td, name + ".fq")
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-i", bam, "-o", out, "-t", str(cores)] + flags, capture_output=True, env=dict(os.environ, TGSF_TIMING="1"))
        dt = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr.decode()[-500:]
        lines = open(out, "rb").read().split(b"\n")
        outs[name] = sorted(b"\n".join(lines[i:i + 4]) for i in range(0, len(lines) - 3, 4))
        print("%-10s wall %.2f s -> %.3f Gbases/s" % (name, dt, bases / dt / 1e9))
        for l in p.stderr.decode().splitlines():
            if l.startswith("TIMING"): print("   ", l)
    print("same records:", outs["reference"] == outs["mi355x"])
