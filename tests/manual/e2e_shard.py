#!/usr/bin/env python3
"""One process per GPU on a 1-GPU box: does the part-file sink lift the single-file bound?  

    tests/manual/e2e_shard.py [n_reads] [ranks ...]        default: 700000 reads (C2's shape, ~63 GB of text), ranks 2 3 4

Runs config C2's command line (automatic pre-pass) on a tmpfs file: one process writing ONE file (the headline way),
then N rank processes SHARING device 0, each writing its own part (tgsfilter --ranks N --devices 0), then both into
/dev/null.  Checks that the parts, concatenated, are the single process's file (sha256) and that the INFO lines agree.
Prints one line per run; the SHARD / TIMING lines of the last run of each kind follow."""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tgsfilter_amd import synth  # noqa: E402

CLI = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")


def sha(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            while True:
                b = f.read(1 << 26)
                if not b:
                    break
                h.update(b)
    return h.hexdigest()


def info(err):
    return [l for l in err.splitlines() if l.startswith("INFO: ") and "written to" not in l]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 700_000
    ranks_list = [int(a) for a in sys.argv[2:]] or [2, 3, 4]
    reps = int(os.environ.get("REPS", "3"))
    td = tempfile.mkdtemp(prefix="tgsf_shard_", dir="/dev/shm")
    try:
        fq = os.path.join(td, "c2.fq")
        t0 = time.perf_counter()
        bases, nbytes = synth.write_ont_fastq(fq, n, seed=2, procs=min(32, (os.cpu_count() or 8)), mean_len=45000.0, max_len=2_000_000)
        print("input: %d reads, %.2f Gbases, %.1f GB of text (%.0f s to write)" % (n, bases / 1e9, nbytes / 1e9, time.perf_counter() - t0), flush=True)
        flags = ["-x", "ont", "-l", "1000", "-q", "10", "-t", "32"]
        env = dict(os.environ, TGSF_TIMING="1")
        ref_sha = ref_info = None

        def run(tag, extra, out, parts=0, check=True):
            nonlocal ref_sha, ref_info
            walls, err = [], ""
            for _ in range(reps):
                for f in ([out] if not parts else ["%s.part%d" % (out, r) for r in range(parts)]):
                    if os.path.isfile(f) and not os.path.islink(f):
                        os.remove(f)
                t1 = time.perf_counter()
                p = subprocess.run([CLI, "-i", fq, "-o", out] + flags + extra, capture_output=True, env=env)
                walls.append(time.perf_counter() - t1)
                err = p.stderr.decode()
                assert p.returncode == 0, err[-3000:]
            same = ""
            if check:
                s = sha([out] if not parts else ["%s.part%d" % (out, r) for r in range(parts)])
                if ref_sha is None:
                    ref_sha, ref_info = s, info(err)
                same = "  output sha256 %s, INFO lines %s" % ("== single file's" if s == ref_sha else "DIFFERS", "same" if info(err) == ref_info else "DIFFER")
                assert s == ref_sha and info(err) == ref_info
            print("%-34s wall %s -> best %.2f s = %.2f Gbases/s, mean %.2f Gbases/s%s" % (
                tag, " ".join("%.2f" % w for w in walls), min(walls), bases / min(walls) / 1e9, bases * len(walls) / sum(walls) / 1e9, same), flush=True)
            return err

        out = os.path.join(td, "out.fq")
        e1 = run("one process, ONE tmpfs file", ["--devices", "0"], out)
        os.remove(out)
        last = {}
        for r in ranks_list:
            last[r] = run("%d ranks on device 0, %d part files" % (r, r), ["--ranks", str(r), "--devices", "0"], out, parts=r)
            for k in range(r):
                os.remove("%s.part%d" % (out, k))
        null = os.path.join(td, "null.fq")
        os.symlink("/dev/null", null)
        run("one process, /dev/null", ["--devices", "0"], null, check=False)
        for r in ranks_list[:2]:
            for k in range(r):
                if not os.path.islink("%s.part%d" % (null, k)):
                    os.symlink("/dev/null", "%s.part%d" % (null, k))
            run("%d ranks on device 0, /dev/null" % r, ["--ranks", str(r), "--devices", "0"], null, parts=r, check=False)
        print("---- one process:")
        for l in e1.splitlines():
            if l.startswith(("TIMING", "GPU:", "RESERVE")):
                print("   ", l)
        for r, e in last.items():
            print("---- %d ranks:" % r)
            for l in e.splitlines():
                if l.startswith(("SHARD", "TIMING")):
                    print("   ", l[:700])
    finally:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
