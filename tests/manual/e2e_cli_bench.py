#!/usr/bin/env python3
"""End-to-end wall time of the command line (FASTQ file in, FASTQ file out, tmpfs) against the reference
binary on the same box: tests/manual/e2e_cli_bench.py [n_reads] [mean_len].  Also checks that both wrote the
same multiset of records (the reference's order is nondeterministic with -t > 1)."""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tgsfilter_amd import synth  # noqa: E402


def gen(path, n, mean_len, seed=3):
    rng = np.random.default_rng(seed)
    lens = synth.ont_lengths(rng, n, mean_len)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    bases = 0
    with open(path, "wb") as f:
        for i in range(n):
            L = int(lens[i])
            s = acgt[rng.integers(0, 4, L)]
            if rng.random() < 0.8:
                a = np.frombuffer(synth.mutate(rng, synth.ONT_RAPID, 0.1), dtype=np.uint8)
                pre = int(rng.integers(0, 31))
                if pre + len(a) < L:
                    s[pre:pre + len(a)] = a
            mq = float(rng.choice([7, 9, 12, 14, 18]))
            q = (np.clip(np.rint(rng.normal(mq, 4, L)), 1, 50) + 33).astype(np.uint8)
            f.write(b"@r%d\n" % i + s.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
            bases += L
    return bases


def digest(path):
    lines = open(path, "rb").read().split(b"\n")
    recs = [b"\n".join(lines[i:i + 4]) for i in range(0, len(lines) - 3, 4)]
    return hashlib.md5(b"\n".join(sorted(recs))).hexdigest(), len(recs)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    mean_len = float(sys.argv[2]) if len(sys.argv) > 2 else 45000.0
    tmp = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=tmp) as td:
        fq = os.path.join(td, "in.fq")
        t0 = time.time()
        bases = gen(fq, n, mean_len)
        print("generated %d reads, %.1f Mbases, %.0f MB FASTQ in %.1f s" % (n, bases / 1e6, os.path.getsize(fq) / 1e6, time.time() - t0))
        fa = os.path.join(td, "ad.fa")
        open(fa, "wb").write(b">rapid\n" + synth.ONT_RAPID + b"\n")
        flags = ["-x", "ont", "-l", "1000", "-q", "10", "-5", "0", "-3", "0", "-a", fa]
        cores = max(1, min((os.cpu_count() or 2) - 1, 32))
        res = {}
        for name, exe in (("reference", os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")),
                          ("mi355x", os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter"))):
            if not os.path.exists(exe):
                print(name, "binary missing")
                continue
            out = os.path.join(td, name + ".fq")
            best = None
            for rep in range(2):
                if os.path.exists(out):
                    os.remove(out)          # truncating the previous run's output (GBs of tmpfs pages) is not part of a run
                t0 = time.perf_counter()
                p = subprocess.run([exe, "-i", fq, "-o", out, "-t", str(cores)] + flags, capture_output=True,
                                   env=dict(os.environ, TGSF_TIMING="1"))
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            res[name] = (best, digest(out), [l for l in p.stderr.decode().splitlines() if l.startswith("INFO: ") and "written to" not in l and "reads with a total" in l])
            print("%-10s wall %.2f s  -> %.3f Gbases/s   records %d" % (name, best, bases / best / 1e9, res[name][1][1]))
            for l in p.stderr.decode().splitlines():
                if l.startswith("TIMING"):
                    print("   ", l)
        if len(res) == 2:
            print("same record multiset:", res["reference"][1] == res["mi355x"][1], " same totals:", res["reference"][2] == res["mi355x"][2])
            print("speed-up %.1fx" % (res["reference"][0] / res["mi355x"][0]))


if __name__ == "__main__":
    main()
