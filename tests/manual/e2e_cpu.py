#!/usr/bin/env python3
"""Where the command line's CPU seconds go, and what a setting does to them (the CPU budget of an N-GPU job).

    tests/manual/e2e_cpu.py [n_reads] [VAR=v1,v2,... ...]      default: 1333334 reads (one of C2's three files)

Writes one C2-shaped FASTQ file to tmpfs, then runs config C2's command line on it REPS (default 2) times per setting --
the baseline first, then every value of every VAR (test settings: TGSF_DEBUG_KNOBS=1 is set) -- and prints per run: wall
seconds, the control group's CPU seconds, and the CPU: line's stages (host/cputime.h)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tgsfilter_amd import synth  # noqa: E402

CLI = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")


def main():
    args = sys.argv[1:]
    n = int(args.pop(0)) if args and args[0].isdigit() else 1_333_334
    # VAR=v1,v2 (one run per value) or A=x+B=y (one run with both set)
    combos = [dict(kv.split("=", 1) for kv in a.split("+")) for a in args if "+" in a]
    sweeps = [(a.split("=", 1)[0], a.split("=", 1)[1].split(",")) for a in args if "+" not in a]
    ranks = int(os.environ.get("RANKS", "0"))
    reps = int(os.environ.get("REPS", "2"))
    print("box: shmem_enabled = %s | cpu.max = %s" % (open("/sys/kernel/mm/transparent_hugepage/shmem_enabled").read().strip(),
                                                     bench.cgroup_limits()["cpus"]), flush=True)
    td = tempfile.mkdtemp(prefix="tgsf_cpu_", dir="/dev/shm")
    try:
        fq, out = os.path.join(td, "c2.fq"), os.path.join(td, "out.fq")
        t0 = time.perf_counter()
        bases, nbytes = synth.write_ont_fastq(fq, n, seed=2, procs=min(32, (os.cpu_count() or 8)), mean_len=45000.0, max_len=2_000_000)
        print("input: %d reads, %.2f Gbases, %.1f GB of text (%.0f s to write)" % (n, bases / 1e9, nbytes / 1e9, time.perf_counter() - t0), flush=True)
        flags = ["-x", "ont", "-l", "1000", "-q", "10", "-t", "32"]

        def run(tag, extra_env, sink=out):
            for _ in range(reps):
                for f in [out] + ["%s.part%d" % (out, r) for r in range(64)]:
                    if os.path.isfile(f) and not os.path.islink(f):
                        os.remove(f)
                env = dict(os.environ, TGSF_TIMING="1", TGSF_DEBUG_KNOBS="1", **extra_env)
                c0 = bench.cgroup_cpu()
                t0 = time.perf_counter()
                p = subprocess.run([CLI, "-i", fq, "-o", sink] + flags + (["--ranks", str(ranks), "--devices", "0"] if ranks else []), capture_output=True, env=env)
                dt = time.perf_counter() - t0
                c1 = bench.cgroup_cpu()
                if p.returncode:
                    print(tag, "FAILED", p.stderr.decode()[-500:])
                    return
                t = bench.parse_timing(p.stderr.decode())
                st = t.get("cpu_s_by_stage", {})
                print("%-28s wall %.3f | cgroup cpu %.1f s (throttled %.1f) | CPU line %.1f s = %.3f CPU-s/Gbase | fallocate wall %.2f, mapping pages wall %.2f | %s"
                      % (tag, dt, c1["usage_s"] - c0["usage_s"] if c0 and c1 else -1, c1["throttled_s"] - c0["throttled_s"] if c0 and c1 else -1,
                         t.get("cpu_s", 0), t.get("cpu_s_per_gbase") or 0, t.get("fallocate_s", 0), t.get("populate_s", 0),
                         ", ".join("%s %.1f" % (re.sub(r" \(.*\)", "", k), v) for k, v in st.items() if v >= 0.3)), flush=True)

        run("baseline", {})
        for var, values in sweeps:
            for v in values:
                run("%s=%s" % (var, v), {var: v})
        for c in combos:
            run("+".join("%s=%s" % (k, os.path.basename(v)) for k, v in c.items()), c)
        null = os.path.join(td, "null.fq")
        os.symlink("/dev/null", null)
        for r in range(ranks):
            os.symlink("/dev/null", "%s.part%d" % (null, r))
        run("baseline -> /dev/null", {}, null)
        for c in combos:
            run("+".join("%s=%s" % (k, os.path.basename(v)) for k, v in c.items()) + " -> /dev/null", c, null)
    finally:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
