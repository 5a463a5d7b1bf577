#!/usr/bin/env python3
"""Bounded-memory check of the streamed-input path on the GPU box: a C2-shaped FASTQ of N reads, gzip'ed as many
members, run through the command line; peak RSS of the run and the output against the plain-file run.
tests/manual/stream_rss.py [n_reads]"""
import os, resource, subprocess, sys, tempfile, time, zlib
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tgsfilter_amd import synth


def comp(job):
    path, off, n = job
    with open(path, "rb") as f:
        f.seek(off)
        co = zlib.compressobj(1, zlib.DEFLATED, 31)
        return co.compress(f.read(n)) + co.flush()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    td = tempfile.mkdtemp(prefix="rss_", dir="/dev/shm")
    fq = os.path.join(td, "in.fq")
    bases, nbytes = synth.write_ont_fastq(fq, n, seed=2)
    piece = 64 << 20
    jobs = [(fq, o, min(piece, nbytes - o)) for o in range(0, nbytes, piece)]
    t0 = time.time()
    with mp.Pool(32) as pool, open(fq + ".gz", "wb") as out:
        for blob in pool.imap(comp, jobs):
            out.write(blob)
    print("text %.2f GB -> gz %.2f GB (%d members) in %.1f s" % (nbytes / 1e9, os.path.getsize(fq + ".gz") / 1e9, len(jobs), time.time() - t0))
    fa = os.path.join(td, "ad.fa"); open(fa, "wb").write(b">rapid\n" + synth.ONT_RAPID + b"\n")
    exe = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
    flags = ["-t", "32", "-x", "ont", "-l", "1000", "-q", "10", "-5", "0", "-3", "0", "-a", fa]
    res = {}
    for tag, inp in (("plain", fq), ("gz", fq + ".gz")):
        out = os.path.join(td, tag + ".fq")
        t0 = time.time()
        p = subprocess.Popen([exe, "-i", inp, "-o", out] + flags, stderr=subprocess.PIPE, env=dict(os.environ, TGSF_TIMING="1"))
        _, status, ru = os.wait4(p.pid, 0)
        err = p.stderr.read().decode()
        dt = time.time() - t0
        assert status == 0, err[-2000:]
        ms = subprocess.run([os.path.join(ROOT, "tools", "fq_multiset"), out], capture_output=True).stdout.decode().split()
        res[tag] = ms
        print("%-5s wall %.2f s, peak RSS %.2f GB, output %s" % (tag, dt, ru.ru_maxrss / 1e6, ms))
        print("     ", [l for l in err.splitlines() if l.startswith("TIMING")][-1:])
    print("same output:", res["plain"] == res["gz"])
    import shutil; shutil.rmtree(td)


if __name__ == "__main__":
    main()
