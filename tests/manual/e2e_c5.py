#!/usr/bin/env python3
"""Config C5's flags end to end (for information; bench.py's headline is C2): the C2 file (ONT reads, lognormal mean
45 kb) with the repeat gate and downsampling switched on, the command line against the reference on the same tmpfs file, for
-k 11 (the default: LDS bitmap kernel) and -k 15 / -k 21 (keys kernel, 32- / 64-bit).  tests/manual/e2e_c5.py [n_reads]
tests/manual/e2e_c5.py <n_reads> c5: config C5 AS WRITTEN instead -- ultra-long reads (lognormal, mean 150 kb, max 2 Mb), the automatic
pre-pass, -g 3g -d 40 -p 100 -k 11 (and -g 1g -d 10, so that the downsampling has something to cut at a size that fits the box)."""
import os, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tgsfilter_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
as_written = len(sys.argv) > 2 and sys.argv[2] == "c5"
td = tempfile.mkdtemp(prefix="c5_", dir="/dev/shm")
fq = os.path.join(td, "c5.fq")
t0 = time.time()
bases, nbytes = synth.write_ont_fastq(fq, n, seed=5, **({"mean_len": 150000.0, "max_len": 2_000_000} if as_written else {}))
print("%d ONT reads%s, %.2f Gbases, %.1f GB of text in %.1f s" % (n, " (lognormal, mean 150 kb, max 2 Mb)" if as_written else "", bases / 1e9, nbytes / 1e9, time.time() - t0))
fa = os.path.join(td, "rapid.fa")
open(fa, "wb").write(b">rapid\n" + synth.ONT_RAPID + b"\n")
cases = (["-p", "100", "-k", "11"], ["-p", "100", "-k", "11", "-r", str(n // 2)], ["-p", "5", "-k", "13"], ["-p", "1", "-k", "16"], ["-p", "1", "-k", "16", "-g", "100m", "-d", "20"])
if as_written:
    cases = (["-g", "3g", "-d", "40", "-p", "100", "-k", "11"], ["-g", "1g", "-d", "10", "-p", "100", "-k", "11"])
for extra in cases:
    flags = (["-x", "ont", "-l", "1000", "-q", "10", "-t", "32"] if as_written else ["-x", "ont", "-l", "1000", "-q", "10", "-5", "0", "-3", "0", "-a", fa, "-t", "32"]) + extra
    print("flags:", " ".join(extra))
    res = {}
    for tag, exe in (("ours", os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")), ("reference", os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref"))):
        out = os.path.join(td, tag + ".fq")
        best = None
        for rep in range(2 if tag == "ours" else 1):
            if os.path.exists(out): os.remove(out)
            t0 = time.perf_counter()
            p = subprocess.run([exe, "-i", fq, "-o", out] + flags, capture_output=True, env=dict(os.environ, TGSF_TIMING="1"))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        ms = subprocess.run([os.path.join(ROOT, "tools", "fq_multiset"), out], capture_output=True).stdout.decode().split()
        info = sorted(l.split(":", 2)[-1] if "input adapter" in l else l for l in p.stderr.decode().splitlines() if l.startswith("INFO: ") and "written to" not in l)
        res[tag] = (ms, info)
        print("%-9s wall %.2f s -> %.2f Gbases/s, output %s" % (tag, best, bases / best / 1e9, ms))
        for l in p.stderr.decode().splitlines():
            if l.startswith("TIMING") or l.startswith("DOWN"): print("   ", l[:1200])
    print("same output multiset:", res["ours"][0] == res["reference"][0], " same INFO lines:", res["ours"][1] == res["reference"][1])
    if res["ours"][1] != res["reference"][1]:
        for a, b in zip(res["ours"][1], res["reference"][1]):
            if a != b: print("   ", a, "|", b)

shutil.rmtree(td)
