#!/usr/bin/env python3
"""Config C3's shape end to end (for information; bench.py's headline is C2): HiFi reads N(18 kb, 3 kb), Q~N(30,6), blunt
adapter at the README's rates, -x hifi -l 1000 -q 20 -M 35 -T 50 with automatic pre-pass, the command line against the
reference on the same tmpfs file.  tests/manual/e2e_hifi.py [n_reads]"""
import os, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tgsfilter_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
td = tempfile.mkdtemp(prefix="hifi_", dir="/dev/shm")
fq = os.path.join(td, "c3.fq")
t0 = time.time()
bases, nbytes = synth.write_ont_fastq(fq, n, seed=3, mean_len=18000.0, kind="hifi", reads_per_job=1024)
print("%d HiFi reads, %.2f Gbases, %.1f GB of text in %.1f s" % (n, bases / 1e9, nbytes / 1e9, time.time() - t0))
flags = ["-x", "hifi", "-l", "1000", "-q", "20", "-M", "35", "-T", "50", "-t", "32"]
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
    info = [l for l in p.stderr.decode().splitlines() if l.startswith("INFO: ") and "written to" not in l]
    res[tag] = (ms, info)
    print("%-9s wall %.2f s -> %.2f Gbases/s, output %s" % (tag, best, bases / best / 1e9, ms))
    for l in p.stderr.decode().splitlines():
        if l.startswith("TIMING") or l.startswith("PREPASS"): print("   ", l[:700])
print("same output multiset:", res["ours"][0] == res["reference"][0], " same INFO lines:", res["ours"][1] == res["reference"][1])
if res["ours"][1] != res["reference"][1]:
    for a, b in zip(res["ours"][1], res["reference"][1]):
        if a != b: print("   ", a, "|", b)
shutil.rmtree(td)
