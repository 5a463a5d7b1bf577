#!/usr/bin/env python3
"""Wall time with -o *.fq.gz (per-record gzip members) against the reference, same box: tests/manual/e2e_gz_check.py [n_reads]"""
import gzip, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "manual"))
import e2e_cli_bench as e  # noqa: E402
from tgsfilter_amd import synth  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    fq = os.path.join(td, "in.fq")
    bases, _ = synth.write_ont_fastq(fq, n, seed=2)
    fa = os.path.join(td, "ad.fa"); open(fa, "wb").write(b">rapid\n" + synth.ONT_RAPID + b"\n")
    flags = ["-x", "ont", "-l", "1000", "-q", "10", "-5", "0", "-3", "0", "-a", fa]
    cores = max(1, min((os.cpu_count() or 2) - 1, 32))
    dig = {}
    for name, exe in (("reference", os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")), ("mi355x", os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter"))):
        out = os.path.join(td, name + ".fq.gz")
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-i", fq, "-o", out, "-t", str(cores)] + flags, capture_output=True)
        dt = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr.decode()[-500:]
        data = gzip.open(out, "rb").read()
        lines = data.split(b"\n")
        recs = sorted(b"\n".join(lines[i:i + 4]) for i in range(0, len(lines) - 3, 4))
        dig[name] = (len(recs), hash(tuple(recs)))
        print("%-10s wall %.2f s -> %.3f Gbases/s  gz size %.0f MB  records %d" % (name, dt, bases / dt / 1e9, os.path.getsize(out) / 1e6, len(recs)))
    print("same records:", dig["reference"] == dig["mi355x"])
