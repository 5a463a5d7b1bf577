#!/usr/bin/env python3
"""Host-pipeline experiments on the GPU box: one C2-shaped FASTQ file on tmpfs, the command line run under several
knob settings, TIMING lines printed.  tests/manual/e2e_knobs.py [n_reads] 'ENV=V ENV2=V' 'ENV=...' ..."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tgsfilter_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
td = tempfile.mkdtemp(prefix="knobs_", dir="/dev/shm")
fq = os.path.join(td, "in.fq")
bases, nbytes = synth.write_ont_fastq(fq, n, seed=2)
fa = os.path.join(td, "ad.fa"); open(fa, "wb").write(b">rapid\n" + synth.ONT_RAPID + b"\n")
exe = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
for setting in (sys.argv[2:] or [""]):
    env = dict(os.environ, TGSF_TIMING="1")
    out = os.path.join(td, "out.fq")
    pre = []
    for kv in setting.split():
        k, v = kv.split("=")
        if k == "OUT": out = v
        elif k == "TASKSET": pre = ["taskset", "-c", v]
        else: env[k] = v
    for rep in range(2):
        if os.path.isfile(out) and not os.path.islink(out): os.remove(out)
        t0 = time.perf_counter(); e0 = time.time()
        p = subprocess.run(pre + [exe, "-i", fq, "-o", out, "-t", "32", "-x", "ont", "-l", "1000", "-q", "10", "-5", "0", "-3", "0", "-a", fa], capture_output=True, env=env)
        dt = time.perf_counter() - t0; e1 = time.time()
    print("[%s] wall %.3f s -> %.2f Gbases/s" % (setting, dt, bases / dt / 1e9))
    for l in p.stderr.decode().splitlines():
        if l.startswith("CLOCK"):
            w = l.replace(",", "").split()
            print("    before main %.3f s, after leaving main %.3f s" % (float(w[4]) - e0, e1 - float(w[7])))
    print("   ", [l for l in p.stderr.decode().splitlines() if l.startswith(("TIMING", "POOL", "EXIT_PROBE"))][-3:] or p.stderr.decode()[-500:])
import shutil; shutil.rmtree(td)
