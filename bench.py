#!/usr/bin/env python3
"""bench.py -- filtered Gbases/s of TGSFilter's per-read filtering path on N x MI355X.

Contract (see the driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run, one rank per GPU.

STDOUT carries ONE line of at most 4 kB (the contract's keys, `roofline`, `cpu_baseline`, the speed-up, `scaling_figures`);
everything else this program measures goes to `--detail-file` (default: bench_detail.json beside this file) and to stderr.

HEADLINE (`value`): the END-TO-END metric of BASELINE.json / SURVEY 8(d) -- input bases / wall seconds of the
whole command line (tgsfilter_amd/bin/tgsfilter) on FASTQ text of config C2's shape held on tmpfs: process start,
library load, index, pre-pass, H2D, kernels, D2H of reads and fragments, formatting, output file written and closed,
report written.  C2's 4 M reads are staged as consecutive files (the box's memory control group does not hold them at
once); a STEP = one run of the command line over ONE staged file, the K timed steps dealt over the files in staging
order (see e2e_leg).  The reference binary (oracle/_ref/tgsfilter_ref -t <cores>) runs on the SAME files with the SAME
flags and the SAME sink in the same bench run: that is `cpu_baseline`; the output files are compared as multisets of
records (the reference's order is nondeterministic with -t > 1) and the INFO lines one by one.  Both sinks SURVEY 8(d)
allows are measured (a tmpfs file -- the headline -- and /dev/null), see `e2e` in the detail file.

`kernel_path`: the device-resident figure (batches already in HBM, tgsf_submit_device), kept separate; the
`roofline` block is measured on it with HIP events on the launch stream inside the library.

N > 1: the end-to-end runs are `tgsfilter --ranks N --devices 0..N-1` (one rank process per GPU, a part file each, one
all-reduce of the tallies; no reference at N > 1); the kernel path runs the FIXED C4 job (31 batches = 4.06 M reads)
dealt over the ranks -- strong scaling -- with ONE all-reduce (RCCL) of the tally vector inside the timed region.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLI = os.environ.get("TGSF_BENCH_CLI") or os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")   # override: tests only
REF = os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")
# How the command line ends (host/main.cpp): by default one process, every mapping and GPU context taken down before it
# returns ("sync"); TGSF_DETACH=1 makes it work in a child whose teardown goes on after the caller has its status.
DEFAULT_EXIT_MODE = "sync"
FQ_MULTISET = os.path.join(ROOT, "tools", "fq_multiset")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_hash():
    """Identifies the kernel sources a profile was taken with (profiles/traffic.json is stamped with it)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "tgsfilter_amd", "csrc")
    for fn in ("tgsf_hip.h", "tgsf_core.h", "tgsf_dev.h", "tgsf_kernels.h", "tgsf_lib.hip"):
        h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------
# end-to-end leg: the command line on a FASTQ file (no GPU is touched by THIS process before it is over)
# ---------------------------------------------------------------------------------------------------------
def info_lines(stderr_text):
    # the set of -a adapters is an unordered_set in the reference: its listing order is not part of the comparison
    lines = [l for l in stderr_text.splitlines() if l.startswith("INFO: ") and "written to" not in l]
    ads = sorted(l.split(":", 2)[2] for l in lines if l.startswith("INFO: input adapter"))
    return ads + [l for l in lines if not l.startswith("INFO: input adapter")]


def multiset(path):
    if not os.path.exists(FQ_MULTISET):
        subprocess.run(["gcc", "-O2", "-pthread", "-o", FQ_MULTISET, FQ_MULTISET + ".c"], check=True)
    return subprocess.run([FQ_MULTISET, path], capture_output=True, check=True).stdout.decode().split()


def cli_processes():
    """PIDs still running the command line under test, INCLUDING one that is on its way out: a process taking down
    hundreds of GB of mappings in exit has no /proc/<pid>/exe any more (its mm is detached first) but keeps its
    name and is not a zombie yet -- and still holds the tmpfs pages of a deleted output file."""
    real = os.path.realpath(CLI)
    comm = os.path.basename(CLI)[:15]
    me = os.getpid()
    out = []
    for d in os.listdir("/proc"):
        if not d.isdigit() or int(d) == me:
            continue
        try:
            stat = open("/proc/%s/stat" % d).read()
            name, state = stat[stat.index("(") + 1:stat.rindex(")")], stat[stat.rindex(")") + 2]
            if state in "ZX":
                # a zombie thread-group LEADER whose other threads are still on their way out: the last of them is the
                # one that takes the address space down
                alive = False
                for t in os.listdir("/proc/%s/task" % d):
                    ts = open("/proc/%s/task/%s/stat" % (d, t)).read()
                    if ts[ts.rindex(")") + 2] not in "ZX":
                        alive = True
                        break
                if not alive:
                    continue
                if name == comm:
                    out.append(int(d))
                continue
            try:
                if os.path.realpath(os.readlink("/proc/%s/exe" % d)) == real:
                    out.append(int(d))
                    continue
            except OSError:
                if name == comm:              # exiting: no exe link any more
                    out.append(int(d))
        except (OSError, ValueError, IndexError):
            pass
    return out


def wait_cli_gone(limit_s=180.0):
    """Nothing of an earlier run may overlap the next timed run (neither ours nor the reference's): wait until no
    process runs the command line any more (with TGSF_DETACH=1 a child takes its mappings down after the parent
    returned).  Returns the seconds waited (not part of any run's wall time)."""
    t0 = time.perf_counter()
    while cli_processes() and time.perf_counter() - t0 < limit_s:
        time.sleep(0.02)
    return time.perf_counter() - t0


def cgroup_cpu():
    """CPU seconds used / throttled so far by this box's control group (cgroup v2), or None."""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
        return {"usage_s": int(d["usage_usec"]) * 1e-6, "throttled_s": int(d.get("throttled_usec", 0)) * 1e-6,
                "nr_throttled": int(d.get("nr_throttled", 0))}
    except (OSError, KeyError, ValueError):
        return None


def cgroup_limits():
    """What the box's control group allows: CPUs' worth of time per second (cpu.max) and bytes of memory (memory.max)
    -- tmpfs pages count as memory of the group that wrote them."""
    out = {"cpus": None, "memory_bytes": None, "memory_used_bytes": None}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            out["cpus"] = int(q) / int(per)
    except (OSError, ValueError):
        pass
    try:
        m = open("/sys/fs/cgroup/memory.max").read().strip()
        if m != "max":
            out["memory_bytes"] = int(m)
        out["memory_used_bytes"] = int(open("/sys/fs/cgroup/memory.current").read().strip())
    except (OSError, ValueError):
        pass
    return out


LAST_RUN_CPU = {}


def run_cmd(cmd, env=None):
    wait_cli_gone()
    c0 = cgroup_cpu()
    t0 = time.perf_counter()
    p = subprocess.run(cmd, capture_output=True, env=env)
    dt = time.perf_counter() - t0
    c1 = cgroup_cpu()
    LAST_RUN_CPU.clear()
    if c0 and c1:
        LAST_RUN_CPU.update({"cpu_s": c1["usage_s"] - c0["usage_s"], "throttled_thread_s": c1["throttled_s"] - c0["throttled_s"],
                             "throttled_periods": c1["nr_throttled"] - c0["nr_throttled"], "wall_s": dt})
    if p.returncode != 0:
        raise SystemExit("command failed (%d): %s\n%s" % (p.returncode, " ".join(cmd), p.stderr.decode()[-3000:]))
    return dt, p.stderr.decode()


def parse_timing(err):
    """Numbers of the command line's TGSF_TIMING lines: where the wall time went."""
    import re
    t = {}
    for l in err.splitlines():
        if l.startswith("TIMING:"):
            for key, pat in (("total_s", r"total ([0-9.]+) s"), ("index_prepass_s", r"index\+prepass ([0-9.]+)"),
                             ("library_wait_s", r"waiting for the library ([0-9.]+)"), ("pipeline_s", r"pipeline ([0-9.]+)"),
                             ("fallocate_s", r"\(fallocate ([0-9.]+)"), ("populate_s", r"mapping the reserved pages ([0-9.]+)"),
                             ("submit_summed_s", r"feeders ([0-9.]+)"), ("release_tail_s", r"dropping the mappings ([0-9.]+)")):
                m = re.search(pat, l)
                if m:
                    t[key] = float(m.group(1))
        elif l.startswith("GPU:"):
            m = re.search(r"kernels ([0-9.]+) s", l)
            if m:
                t["gpu_kernel_s_summed"] = float(m.group(1))
        elif l.startswith("CPU:"):
            # CPU seconds by stage (host/cputime.h): thread CPU time, user + system (a sharded job prints one line per rank: summed)
            m = re.match(r"CPU: ([0-9.]+) s of CPU time \(user \+ system\) for ([0-9.]+) Gbases", l)
            if m:
                t["cpu_s"] = t.get("cpu_s", 0.0) + float(m.group(1))
                t["cpu_gbases"] = t.get("cpu_gbases", 0.0) + float(m.group(2))
                st = t.setdefault("cpu_s_by_stage", {})
                for part in l.split("|")[1].split(","):
                    mm = re.match(r"\s*(.+?) ([0-9.]+)\s*$", part)
                    if mm:
                        st[mm.group(1)] = round(st.get(mm.group(1), 0.0) + float(mm.group(2)), 3)
                mm = re.search(r"threads of the runtime and others (-?[0-9.]+)", l)
                if mm:
                    st["runtime threads and others"] = round(st.get("runtime threads and others", 0.0) + float(mm.group(1)), 3)
                t["cpu_s_per_gbase"] = t["cpu_s"] / t["cpu_gbases"] if t["cpu_gbases"] else None
    return t


class Budget:
    """The driver gives a bench run a time limit: optional legs are dropped (and named) when the time runs short;
    the K timed steps of the headline never are."""
    def __init__(self, limit_s):
        self.t0, self.limit, self.skipped = time.perf_counter(), limit_s, []

    def left(self):
        return self.limit - (time.perf_counter() - self.t0)

    def allows(self, name, need_s):
        if self.left() >= need_s:
            return True
        self.skipped.append("%s (needs ~%.0f s, %.0f s left of --e2e-budget-s)" % (name, need_s, self.left()))
        log("bench: skipping %s" % self.skipped[-1])
        return False


# The BASELINE.json configurations the end-to-end leg can run (`--config`).  C5 as written: ONT ultra-long reads (lognormal,
# mean 150 kb, max 2 Mb) through the repeat gate (-p/-k; `-k` only acts with `-p` > 0, src/TGSFilter.cpp:1982: -p 100 is
# this bench's choice, said in SURVEY 8d) and the longest-first downsampling (-g 3g -d 40).  The reference keeps a copy of
# the filtered reads (<inprefix>.tmp.XXXXX.fq, :3129-3137) beside its output: three files on tmpfs at once.
REF_RUNS = 1                                  # runs of the reference per thread count on the first file (round 5 ran 2 x 2: -t 32 and
                                              # -t 15 differ by < 1 % on this pool's boxes, the question is settled)
E2E_CONFIGS = {
    "c2": {"name": "C2 (BASELINE.json configs[1])", "reads": 4_000_000, "mean_len": 45000.0, "max_len": 2_000_000, "per_read": 90_300, "files": 2.0,
           "flags": ["-x", "ont", "-l", "1000", "-q", "10"], "seed": 2, "what": "automatic trims and adapter identification", "split": True},
    "c3": {"name": "C3 (BASELINE.json configs[2]) at one GPU", "reads": 19_000_000, "mean_len": 18000.0, "max_len": 40_000, "per_read": 36_200, "files": 2.0,
           "flags": ["-x", "hifi", "-l", "1000", "-q", "20", "-M", "35", "-T", "50"], "seed": 3, "kind": "hifi", "reads_per_job": 1024,
           "what": "HiFi reads N(18 kb, 3 kb), PacBio blunt adapter at the README's rates, automatic pre-pass, middle-adapter split"},
    "c5": {"name": "C5 (BASELINE.json configs[4]) at one GPU", "reads": 1_000_000, "mean_len": 150000.0, "max_len": 2_000_000, "per_read": 300_100, "files": 3.0,
           "flags": ["-x", "ont", "-l", "1000", "-q", "10", "-g", "3g", "-d", "40", "-p", "100", "-k", "11"], "seed": 5,
           "what": "automatic pre-pass, repeat gate -p 100 -k 11 on the GPU, longest-first downsampling to 3 Gb x 40"},
}


def cpu_budget():
    """CPUs' worth of time this box's control group allows (cpu.max), or the hardware's threads."""
    lim = cgroup_limits()
    hw = os.cpu_count() or 2
    return int(min(hw, max(1, round(lim["cpus"])))) if lim["cpus"] else hw


def multiset_of(paths):
    """The order-independent digest of the records of several files taken together (the parts of a sharded run):
    [records, sum of record hashes mod 2^64, xor of record hashes, bytes]."""
    tot = [0, 0, 0, 0]
    for p in paths:
        m = [int(x) for x in multiset(p)]
        tot = [tot[0] + m[0], (tot[1] + m[1]) & (2**64 - 1), tot[2] ^ m[2], tot[3] + m[3]]
    return [str(x) for x in tot]


def e2e_leg(args, n_gpus):
    """The command line end to end on the configuration's reads (default: C2's 4 M), staged on tmpfs as ONE file when the
    box's memory holds input + output, as CONSECUTIVE files (same generator, seeds in a row) otherwise -- C2 as written is
    360 GB of text + 215 GB of output, more than the box's memory control group allows at once.
    A STEP = one run of the command line over ONE staged file (C2 on this pool: a third of the 4 M reads, 60 Gbases, 120 GB
    of text); the K timed steps are dealt over the files in staging order (K = 20, 3 files: 7 + 7 + 6), so that K steps pass
    over all the configuration's reads as long as K >= the number of files (fewer steps stage fewer files, and the line says
    so).  value = bases of the K timed steps / their summed wall time.  (Rounds 4-5 called a pass over ALL files a step: 3 K
    runs, each preceded by the removal of the previous 72-GB output -- 5 s of tmpfs page freeing outside any timed region --
    which took the driver's run to 1 450 s of its 1 800-s limit.)
    Per file, in this order:
      ours     W (first file: the others 1) warm-up runs + its share of the K timed runs into a tmpfs file (N > 1:
               `--ranks N`, one process per GPU, a part file each);
      theirs   the reference binary on the same file, same flags, same sink: on the first file at its own thread clamp
               (-t min(hw-1, 32), src/TGSFilter.cpp:488-499) AND at what the box's CPU quota lets run unthrottled
               (-t min(cpu.max - 1, 32)), REF_RUNS run(s) each; on the other files once, at the faster of the two.  Output
               multisets and INFO lines compared with ours for every file;
    then, on the first file only and budget permitting: the /dev/null sink, and (N = 1) the same run as 3 rank processes
    sharing the GPU with a part file each (`sharded`: the program path of N GPUs on this box)."""
    from tgsfilter_amd import synth
    if not os.path.exists(CLI):
        raise SystemExit("bench.py: %s is missing -- run __graft_entry__.build() (there is no fallback path)" % CLI)
    budget = Budget(float(getattr(args, "e2e_budget_s", 1500.0)))
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    if shm and shutil.disk_usage(shm).free < (1 << 30) and shutil.disk_usage(tempfile.gettempdir()).free > shutil.disk_usage(shm).free:
        shm = None                                    # a token /dev/shm (container default of 64 MB): stage on the temporary directory
    cfg = E2E_CONFIGS[getattr(args, "config", "c2") or "c2"]
    n_want = n_reads = args.e2e_reads if getattr(args, "e2e_reads", None) else cfg["reads"]
    per_read = cfg["per_read"]                        # bytes of text per read (2 x mean length + header)
    # Room for the staging: the input and ONE output at a time live on tmpfs (ours is digested and removed before the
    # reference writes its own); an output is never larger than its input.  tmpfs pages are memory of the box's control
    # group: a group that outgrows memory.max loses the whole box, so stay well inside it.
    lim = cgroup_limits()
    free = shutil.disk_usage(shm or tempfile.gettempdir()).free
    why = "free space on %s" % (shm or tempfile.gettempdir())
    if shm and lim["memory_bytes"]:
        room = lim["memory_bytes"] - (lim["memory_used_bytes"] or 0)
        if room < free:
            free, why = room, "the box's memory control group (memory.max %.0f GiB; tmpfs pages count)" % (lim["memory_bytes"] / 2**30)
    try:
        avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
        if shm and avail < free:
            free, why = avail, "available memory"
    except (OSError, IndexError, ValueError):
        pass
    need = n_reads * per_read * cfg["files"]
    n_files = int(getattr(args, "e2e_files", 0) or 0)
    reduced = None
    if not n_files:
        n_files = 1
        if need > 0.85 * free:
            if cfg.get("split"):
                n_files = int(-(-need // (0.80 * free)))
            else:                                     # (a downsampling configuration selects among ALL its reads: one file, fewer reads)
                n_reads = max(2000, int(n_reads * 0.85 * free / need) // 1000 * 1000 or 2000)
                reduced = "%s holds %.0f GB: the file is %d of the configuration's %d reads (input + %s, <= %.0f x %.0f GB, kept under 85 %% of that)" % (
                    why, free / 1e9, n_reads, n_want, "one output" if cfg["files"] == 2.0 else "the reference's temporary copy and its output",
                    cfg["files"], n_reads * per_read / 1e9)
                log("bench: " + reduced)
    per_file = [n_reads // n_files + (1 if f < n_reads % n_files else 0) for f in range(n_files)]
    # the K timed steps dealt over the files (a step = one run over one file); fewer steps than files stage fewer files
    n_staged = max(1, min(n_files, args.steps))
    steps_of = [args.steps // n_staged + (1 if f < args.steps % n_staged else 0) for f in range(n_staged)]
    phase = {"generate": 0.0, "ours": 0.0, "reference": 0.0, "remove_outputs": 0.0, "digests": 0.0}     # where the leg's own time goes
    split_note = None
    if n_files > 1:
        split_note = "%s holds %.0f GB, input + one output of all %d reads need %.0f GB: the reads are staged as %d consecutive files of %s reads (same generator, seeds %s)" % (
            why, free / 1e9, n_reads, need / 1e9, n_files, "/".join(str(x) for x in sorted(set(per_file), reverse=True)),
            ", ".join(str(cfg["seed"] + 1000 * f) for f in range(n_files)))
        log("bench: " + split_note)
    td = tempfile.mkdtemp(prefix="tgsf_bench_", dir=shm)
    hw_cores = max(1, min((os.cpu_count() or 2) - 1, 32))      # the reference clamps -t to min(hw-1, 32), :488-499
    quota_cores = max(1, min(cpu_budget() - 1, 32))            # ... what the box's CPU quota lets run without throttling
    gen_procs = max(1, min((os.cpu_count() or 2) - 1, int(2 * (lim["cpus"] or 16)), 64))
    devs = ",".join(str(d) for d in range(n_gpus))
    env = dict(os.environ, TGSF_TIMING="1")
    env.pop("TGSF_SYNC_EXIT", None)
    env.pop("TGSF_DETACH", None)
    have_ref = os.path.exists(REF) and not args.no_cpu_baseline
    want_ranks = int(getattr(args, "e2e_ranks", 0) or 0) or (n_gpus if n_gpus > 1 else 0)
    rank_devs = devs if n_gpus > 1 else "0"


    def rm(path):
        t0 = time.perf_counter()
        for p in [path, path + ".parts"] + ["%s.part%d" % (path, r) for r in range(64)]:
            if os.path.isfile(p) and not os.path.islink(p):
                os.remove(p)          # dropping a previous run's GBs of tmpfs pages is not part of a run
        phase["remove_outputs"] += time.perf_counter() - t0

    def digest(paths):
        t0 = time.perf_counter()
        m = multiset_of(paths)
        phase["digests"] += time.perf_counter() - t0
        return m

    def ours(fq, out, flags, ranks=0, rdevs=None):
        rm(out)
        how = ["--ranks", str(ranks), "--devices", rdevs or rank_devs] if ranks else ["--devices", devs]
        r = run_cmd([CLI, "-i", fq, "-o", out, "-t", str(hw_cores)] + how + flags, env)
        phase["ours"] += r[0]
        return r

    def theirs(fq, out, flags, t):
        rm(out)
        r = run_cmd([REF, "-i", fq, "-o", out, "-t", str(t)] + flags)
        phase["reference"] += r[0]
        return r

    def timed(fq, out, flags, warm, k, ranks=0, rdevs=None):
        for _ in range(warm):
            ours(fq, out, flags, ranks, rdevs)
        walls, err, cpu = [], "", []
        for _ in range(k):
            dt, err = ours(fq, out, flags, ranks, rdevs)
            walls.append(dt)
            cpu.append(dict(LAST_RUN_CPU))
        return walls, err, cpu

    def out_files(out, ranks):
        return ["%s.part%d" % (out, r) for r in range(ranks)] if ranks else [out]

    def describe(walls, err, bases):
        s = {"runs": len(walls), "exit_mode": DEFAULT_EXIT_MODE, "wall_s": walls, "wall_s_mean": sum(walls) / len(walls),
             "gbases_per_s": bases * len(walls) / sum(walls) / 1e9,
             "timing_line": [l for l in err.splitlines() if l.startswith("TIMING")][-1:], "timing": parse_timing(err)}
        t = s["timing"]
        if t.get("total_s"):
            parts = {"fallocate (tmpfs page instantiation, one thread per file, inode lock)": t.get("fallocate_s", 0.0),
                     "mapping the reserved pages": t.get("populate_s", 0.0),
                     "index + pre-pass": t.get("index_prepass_s", 0.0), "waiting for the library/device": t.get("library_wait_s", 0.0)}
            top = max(parts, key=parts.get)
            s["bound"] = "%s: %.2f s of %.2f s" % (top, parts[top], t["total_s"])
            if "gpu_kernel_s_summed" in t:
                s["gpu_busy_frac"] = t["gpu_kernel_s_summed"] / t["total_s"]      # upper bound: contexts overlap
        # (the ranks share one stderr: a line is one write in the program, but take it wherever it starts)
        shard = re.findall(r"SHARD \d+/\d+: [^\n]*", err)
        if shard:
            s["shard_lines"] = [l[:400] for l in shard]
        return s

    files, sinks, runs_by_threads = [], {}, {}
    ref_t = None                                       # the thread count the reference is timed at on files after the first
    est = {"gen_s": 0.0, "run_s": 0.0, "ref_s": 0.0, "rm_s": 0.0}    # measured on the first file: what every further file will cost

    def reserve(f):
        """Seconds the files after file f still need for what is never dropped (generation, warm-up, the K timed runs,
        the reference once with its digest): optional legs run only while that much stays in the budget."""
        return sum(est["gen_s"] * 1.1 + (min(args.warmup, 1) + steps_of[g]) * (est["run_s"] * 1.1 + est["rm_s"]) + (est["ref_s"] * 1.2 + 30 if have_ref else 0)
                   for g in range(f + 1, n_staged))
    try:
        flags = list(cfg["flags"])                               # the configuration as BASELINE.json writes it
        for f in range(n_staged):
            first = f == 0
            fq = os.path.join(td, "in%d.fq" % f)
            t0 = time.perf_counter()
            bases, nbytes = synth.write_ont_fastq(fq, per_file[f], seed=cfg["seed"] + 1000 * f, procs=gen_procs, mean_len=cfg["mean_len"], max_len=cfg["max_len"],
                                                  kind=cfg.get("kind", "ont"), reads_per_job=cfg.get("reads_per_job", 256))
            phase["generate"] += time.perf_counter() - t0
            if first:
                est["gen_s"] = time.perf_counter() - t0
            log("bench: file %d of %d: %d reads / %.2f Gbases / %.2f GB of FASTQ text written to %s in %.1f s by %d processes"
                % (f + 1, n_staged, per_file[f], bases / 1e9, nbytes / 1e9, fq, time.perf_counter() - t0, gen_procs))
            out_o, out_r = os.path.join(td, "ours.fq"), os.path.join(td, "ref.fq")
            rm0 = phase["remove_outputs"]
            walls, err, cpu = timed(fq, out_o, flags, args.warmup if first else min(args.warmup, 1), steps_of[f], want_ranks)
            s = describe(walls, err, bases)
            if first:
                est["run_s"] = s["wall_s_mean"]
                est["rm_s"] = (phase["remove_outputs"] - rm0) / max(1, args.warmup + steps_of[f] - 1)
            s.update({"file": f, "reads": per_file[f], "bases": bases, "fastq_bytes": nbytes, "cpu_last_run": cpu[-1] if cpu else None})
            if want_ranks:
                s["ranks"] = want_ranks
            info = info_lines(err)
            mine = digest(out_files(out_o, want_ranks)) if have_ref else None
            if mine:
                s["output_records"], s["output_bytes"] = int(mine[0]), int(mine[3])
            rm(out_o)
            if have_ref:
                # the reference on the same file, flags and sink.  First file: both thread counts, REF_RUNS runs each, as
                # the budget allows (the first run's output and INFO lines are compared with ours); later files: once.
                plan = [hw_cores] + ([quota_cores] if quota_cores != hw_cores else []) if first else [ref_t]
                compared = False
                walls = {t: [] for t in plan}
                cpus = {t: [] for t in plan}
                # (every thread count once before any of them a second time: should the budget run short, the comparison of
                # the two counts is what stays)
                for rr in range(REF_RUNS if first else 1):
                    for t in plan:
                        known = [w for v in walls.values() for w in v]
                        if known and first and not budget.allows("reference -t %d run %d on file %d" % (t, rr + 1, f + 1), 1.15 * max(known) + 5 + reserve(f)):
                            continue
                        dt, rerr = theirs(fq, out_r, flags, t)
                        walls[t].append(dt)
                        if first:
                            est["ref_s"] = max(est["ref_s"], dt)
                        cpus[t].append(dict(LAST_RUN_CPU))
                        if not compared:
                            compared = True
                            s["same_counters"] = info_lines(rerr) == info
                            if not s["same_counters"]:
                                raise SystemExit("bench: INFO lines differ from the reference's:\n%s\n---\n%s" % ("\n".join(info), "\n".join(info_lines(rerr))))
                            theirs_ms = digest([out_r])
                            s["same_output_multiset"] = mine == theirs_ms
                            if mine != theirs_ms:
                                raise SystemExit("bench: output differs from the reference's (records sum xor bytes): %s vs %s" % (mine, theirs_ms))
                        rm(out_r)
                for t in plan:
                    rw, rc = walls[t], cpus[t]
                    if not rw:
                        continue
                    if first:
                        runs_by_threads[str(t)] = {"threads": t, "wall_s_runs": rw, "wall_s_mean": sum(rw) / len(rw), "wall_s_min": min(rw),
                                                   "gbases_per_s": bases / (sum(rw) / len(rw)) / 1e9, "gbases_per_s_best_run": bases / min(rw) / 1e9, "cpu_runs": rc,
                                                   "why": "the reference's own clamp, -t min(hw-1, 32)" if t == hw_cores else
                                                          "what the box's CPU quota (cpu.max = %s CPUs) runs without throttling" % lim["cpus"]}
                    else:
                        s["reference_wall_s"], s["reference_threads"], s["reference_cpu"] = rw[0], t, rc[0]
                if first and runs_by_threads:
                    best = max(runs_by_threads.values(), key=lambda v: v["gbases_per_s"])
                    ref_t = best["threads"]
                    s["reference_wall_s"], s["reference_threads"] = best["wall_s_mean"], ref_t
            s["_info"] = info
            s["info_prepass"] = [l for l in info if any(w in l for w in ("trim 5'", "trim 3'", "5' adapter", "3' adapter", "min Phred"))]
            files.append(s)
            log("bench: e2e file %d: %s" % (f + 1, json.dumps({k2: v for k2, v in s.items() if k2 not in ("timing_line", "_info", "shard_lines")})))
            if first:
                # /dev/null through a symlink (the suffix decides the format): no page instantiation, PCIe-bound at one GPU
                null_out = os.path.join(td, "null.fq")
                os.symlink("/dev/null", null_out)
                for r in range(want_ranks):
                    os.symlink("/dev/null", "%s.part%d" % (null_out, r))
                k_null = min(args.steps, 3)
                if budget.allows("dev_null sink", (k_null + 1) * s["wall_s_mean"] + 5 + reserve(f)):
                    w2, e2, _ = timed(fq, null_out, flags, min(args.warmup, 1), k_null, want_ranks)
                    d = describe(w2, e2, bases)
                    d["same_counters_as_the_file_run"] = info_lines(e2) == info
                    if not d["same_counters_as_the_file_run"]:
                        raise SystemExit("bench: INFO lines of the /dev/null run differ from the file run's")
                    sinks["dev_null"] = d
                    log("bench: e2e dev_null: %s" % json.dumps({k2: v for k2, v in d.items() if k2 not in ("timing_line", "shard_lines")}))
                # N = 1: the program path of N GPUs on this box -- rank processes sharing the GPU, a part file each (the tallies
                # go over the ranks' sockets: RCCL refuses two ranks on one device).  Same records, same INFO lines.
                if n_gpus == 1 and not want_ranks and getattr(args, "sharded_leg", True):
                    k_sh = min(args.steps, 3)
                    for nr in (3,):
                        if not budget.allows("sharded leg, %d ranks on one GPU" % nr, (k_sh + 1) * s["wall_s_mean"] + 10 + reserve(f)):
                            break
                        w3, e3, _ = timed(fq, out_o, flags, 1, k_sh, nr, "0")
                        d = describe(w3, e3, bases)
                        d["ranks"], d["devices"] = nr, "0 (shared by all ranks)"
                        d["same_counters_as_the_file_run"] = info_lines(e3) == info
                        if mine:
                            d["same_output_multiset"] = digest(out_files(out_o, nr)) == mine
                        rm(out_o)
                        if not d["same_counters_as_the_file_run"] or d.get("same_output_multiset") is False:
                            raise SystemExit("bench: the sharded run's records or INFO lines differ from the single process's")
                        sinks["tmpfs_part_files_%d_ranks_one_gpu" % nr] = d
                        log("bench: e2e sharded: %s" % json.dumps({k2: v for k2, v in d.items() if k2 not in ("timing_line", "shard_lines")}))
            os.remove(fq)
        # ---- the whole workload: a step = one pass over all the files ----
        step_walls = [w for s in files for w in s["wall_s"]]          # the K timed steps, in the order they ran
        K = len(step_walls)
        tot_bases = sum(s["bases"] for s in files)                     # one pass over the staged files (the reference's work)
        step_bases = sum(s["bases"] * len(s["wall_s"]) for s in files)  # what the K steps filtered
        agg = {"runs": K, "exit_mode": DEFAULT_EXIT_MODE, "wall_s": step_walls, "wall_s_mean": sum(step_walls) / K,
               "gbases_per_s": step_bases / sum(step_walls) / 1e9, "files": n_staged, "steps_per_file": steps_of, "bases_per_step_mean": step_bases / K,
               "per_file": [{k2: v for k2, v in s.items() if k2 != "_info"} for s in files],
               "timing": files[0]["timing"], "timing_line": files[0]["timing_line"], "bound": files[0].get("bound"),
               "gpu_busy_frac": files[0].get("gpu_busy_frac"), "cpu_last_run": files[0].get("cpu_last_run"),
               "info_prepass": files[0]["info_prepass"],
               "output_bytes": sum(s.get("output_bytes", 0) for s in files), "output_records": sum(s.get("output_records", 0) for s in files)}
        if want_ranks:
            agg["ranks"] = want_ranks
        if have_ref and all("reference_wall_s" in s for s in files):
            ref_total = sum(s["reference_wall_s"] for s in files)
            agg.update({"same_counters": all(s.get("same_counters") for s in files), "same_output_multiset": all(s.get("same_output_multiset") for s in files),
                        "reference_wall_s": ref_total, "reference_threads": ref_t, "reference_gbases_per_s": tot_bases / ref_total / 1e9,
                        "reference_runs_by_threads_first_file": runs_by_threads,
                        "speedup_vs_reference": agg["gbases_per_s"] / (tot_bases / ref_total / 1e9),
                        "speedup_note": "ours: bases / wall time of %d timed runs over %d file(s) / the reference at -t %d, the FASTER of the thread counts tried on the first file (%s), "
                                        "every file once" % (K, n_staged, ref_t, ", ".join("-t %s: %.3f Gbases/s" % (k2, v["gbases_per_s"]) for k2, v in runs_by_threads.items()))})
        elif have_ref:
            agg["reference_note"] = "the reference was not timed on every file (budget): no whole-workload baseline"
        sinks["tmpfs_file"] = agg
        for k2 in ("dev_null",) + tuple(k3 for k3 in sinks if k3.startswith("tmpfs_part_files")):
            if k2 in sinks and "reference_wall_s" in files[0]:
                sinks[k2]["speedup_vs_reference_file_run_first_file"] = files[0]["reference_wall_s"] / sinks[k2]["wall_s_mean"]
        res = {"config": cfg["name"], "config_what": cfg["what"], "box": {"cgroup_cpus": lim["cpus"], "cgroup_memory_gib": (lim["memory_bytes"] or 0) / 2**30 or None, "hw_threads": os.cpu_count()},
               "reads": sum(per_file[:n_staged]), "reads_of_config_split": n_reads, "bases": tot_bases, "fastq_bytes": sum(s["fastq_bytes"] for s in files),
               "files": n_staged, "files_of_config_split": n_files, "reads_per_file": per_file[:n_staged], "steps_per_file": steps_of,
               "flags": " ".join(flags), "threads": hw_cores, "devices": devs, "ranks": want_ranks or None,
               "reduced": reduced, "split": split_note,
               "staging": "synthetic FASTQ text written to tmpfs (%s) by tgsfilter_amd/synth.write_ont_fastq before timing; "
               "read by both programs through the page cache" % (shm or "tmp"), "sinks": sinks}
        # the pinned pre-pass on round 2's file, the reference beside it (opt-in since round 5: --pinned-variant)
        n_pin = min(400_000, n_reads)
        if cfg is E2E_CONFIGS["c2"] and getattr(args, "pinned_variant", False) and budget.allows("pinned-pre-pass variant", 45 + n_pin * 1.1e-4):
            fq2 = os.path.join(td, "c2_400k.fq")
            bases2, nbytes2 = synth.write_ont_fastq(fq2, n_pin, seed=2, procs=gen_procs)
            fa = os.path.join(td, "rapid.fa")
            open(fa, "wb").write(b">rapid\n" + synth.ONT_RAPID + b"\n")
            pflags = ["-x", "ont", "-l", "1000", "-q", "10", "-5", "0", "-3", "0", "-a", fa]
            out_o, out_r = os.path.join(td, "pin_ours.fq"), os.path.join(td, "pin_ref.fq")
            w4, e4, _ = timed(fq2, out_o, pflags, 1, 3)
            v = describe(w4, e4, bases2)
            if have_ref:
                m4 = multiset(out_o)
                rm(out_o)
                dt, rerr = theirs(fq2, out_r, pflags, ref_t or hw_cores)
                v.update({"same_counters": info_lines(rerr) == info_lines(e4), "same_output_multiset": multiset(out_r) == m4,
                          "reference_wall_s": dt, "speedup_vs_reference": dt / v["wall_s_mean"]})
                rm(out_r)
            v.update({"reads": n_pin, "bases": bases2, "fastq_bytes": nbytes2, "flags": " ".join(pflags[:-1]) + " rapid.fa"})
            res["variants"] = {"pinned_prepass": v}
        res["skipped"] = budget.skipped
        res["seconds"] = time.perf_counter() - budget.t0
        res["seconds_by_phase"] = {k2: round(v, 1) for k2, v in phase.items()}
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return res


# ---------------------------------------------------------------------------------------------------------
# kernel path: batches resident in HBM
# ---------------------------------------------------------------------------------------------------------
def gen_batch(torch, device, n_reads, seed, mean_len, max_len, workload="ont", hifi_rates=(0.0027, 0.0026, 0.00002)):
    """Synthetic C2 (ont) / C3 (hifi) batch built directly in HBM (generation is outside every timed region)."""
    from tgsfilter_amd import synth
    rng = np.random.default_rng(seed)
    hifi = workload == "hifi"
    if hifi:
        lens = np.minimum(synth.hifi_lengths(rng, n_reads, mean_len, sd=mean_len / 6.0), max_len).astype(np.int64)
    else:
        lens = np.minimum(synth.ont_lengths(rng, n_reads, mean_len), max_len).astype(np.int64)
    padded = (lens + 15) // 16 * 16
    offsets = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(padded, out=offsets[1:])
    total = int(offsets[-1])
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    seq = torch.empty(total, dtype=torch.uint8, device=device)
    qual = torch.empty(total, dtype=torch.uint8, device=device)
    mq_set = np.array([30], dtype=np.float32) if hifi else np.array([7, 9, 12, 14, 18], dtype=np.float32)
    mq = torch.from_numpy(rng.choice(mq_set, n_reads)).to(device)
    per_base_mq = torch.repeat_interleave(mq.to(torch.float16), torch.from_numpy(padded).to(device))
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    CH = 1 << 26
    for s in range(0, total, CH):
        e = min(total, s + CH)
        codes = torch.randint(0, 4, (e - s,), device=device, generator=g)
        seq[s:e] = lut[codes]
        q = torch.randn(e - s, device=device, generator=g) * (6.0 if hifi else 4.0) + per_base_mq[s:e].float()
        qual[s:e] = (q.round().clamp_(2 if hifi else 1, 60 if hifi else 50) + 33).to(torch.uint8)
    del per_base_mq
    # rapid adapter at the 5' end of 80 % of reads (0-30 random bases before it, 10 % errors); 0.03 % middle
    idx, val = [], []
    for i in range(n_reads):
        L = int(lens[i])
        if hifi:
            # blunt adapter 5' in 0.27 %, 3' in 0.26 %, middle in 0.002 % of reads, 3 % errors (README ratios)
            u = rng.random()
            r5, r3, rm = hifi_rates
            if u < r5 + r3 + rm and L > 2000:
                a = synth.mutate(rng, synth.PACBIO_BLUNT, 0.03)
                p = 0 if u < r5 else (L - len(a) if u < r5 + r3 else int(rng.integers(300, L - 300 - len(a))))
                idx.append(np.arange(len(a), dtype=np.int64) + offsets[i] + p)
                val.append(np.frombuffer(a, dtype=np.uint8))
            continue
        if rng.random() < 0.80:
            a = synth.mutate(rng, synth.ONT_RAPID, 0.10)
            pre = int(rng.integers(0, 31))
            if pre + len(a) < L:
                idx.append(np.arange(len(a), dtype=np.int64) + offsets[i] + pre)
                val.append(np.frombuffer(a, dtype=np.uint8))
        if rng.random() < 0.0003 and L > 2000:
            a = synth.mutate(rng, synth.ONT_RAPID if rng.random() < 0.5 else synth.ONT_RAPID_RC, 0.05)
            p = int(rng.integers(300, L - 300 - len(a)))
            idx.append(np.arange(len(a), dtype=np.int64) + offsets[i] + p)
            val.append(np.frombuffer(a, dtype=np.uint8))
    if idx:
        seq[torch.from_numpy(np.concatenate(idx)).to(device)] = torch.from_numpy(np.concatenate(val).copy()).to(device)
    return dict(seq=seq, qual=qual, offsets=torch.from_numpy(offsets[:-1].astype(np.uint64).view(np.int64).copy()).to(device),
                lengths=torch.from_numpy(lens.astype(np.uint32).view(np.int32).copy()).to(device),
                n=n_reads, n_bytes=total, bases=int(lens.sum()), h_lens=lens, h_offsets=offsets)


def oracle_slice_check(torch, batch, p_kwargs, workload, reads_rec, frags_rec, m=256, seed=11, must_include=None):
    """A random slice of the batch through the oracle (the checker, never the thing measured): the per-read records
    and the fragments the HIP path produced for those reads must be identical."""
    from oracle import orc
    from tgsfilter_amd import abi
    rng = np.random.default_rng(seed)
    pick = rng.choice(batch["n"], size=min(m, batch["n"]), replace=False)
    if must_include is not None:
        pick = np.concatenate([pick, np.asarray(must_include, dtype=pick.dtype)])
    pick = np.unique(pick)
    lens = batch["h_lens"][pick]
    offs = np.zeros(len(pick) + 1, dtype=np.int64)
    np.cumsum((lens + 15) // 16 * 16, out=offs[1:])
    seq = np.zeros(int(offs[-1]), dtype=np.uint8)
    qual = np.zeros(int(offs[-1]), dtype=np.uint8)
    for j, i in enumerate(pick):
        o, L = int(batch["h_offsets"][i]), int(lens[j])
        seq[offs[j]:offs[j] + L] = batch["seq"][o:o + L].cpu().numpy()
        qual[offs[j]:offs[j] + L] = batch["qual"][o:o + L].cpu().numpy()
    p = abi.make_params(workload, max_read_len=int(lens.max()), **p_kwargs)
    er, ef, _ = orc.filter_batch(p, seq, qual, offs[:-1].astype(np.uint64), lens.astype(np.uint32))
    got = reads_rec[pick]
    for name in ("sum_q", "flags", "n_frags", "trimmed"):
        assert np.array_equal(got[name], er[name]), "oracle slice: read field %s differs" % name
    k = 0
    for j in range(len(pick)):
        fb, nf = int(got["frag_begin"][j]), int(got["n_frags"][j])
        for name in ("start", "len", "flags", "sum_q"):
            assert np.array_equal(frags_rec[name][fb:fb + nf], ef[name][k:k + nf]), "oracle slice: fragment field %s differs" % name
        k += nf
    return len(pick), int(lens.sum())


def kernel_leg(args, torch, dist, world, rank, local_rank, device, xdev, host_group):
    from tgsfilter_amd import abi, capi, rccl, synth
    from tgsfilter_amd import dist as tdist
    hifi = args.workload == "hifi"
    mean_len = args.mean_len or (18000.0 if hifi else 45000.0)
    # the trims the end-to-end pre-pass resolves on this shape (C2's file: 5' 79, 3' 0; HiFi data: 7 / 8, the reference's
    # README) -- with any trim no read is "kept whole", and the clean tables cost a pass of their own
    head_trim = args.head_trim if args.head_trim is not None else (7 if hifi else 79)
    tail_trim = args.tail_trim if args.tail_trim is not None else (8 if hifi else 0)
    flags = ("-x hifi -l 1000 -q 20 -5 %d -3 %d" if hifi else "-x ont -l 1000 -q 10 -5 %d -3 %d") % (head_trim, tail_trim)
    wl_adapters = [synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC] if hifi else [synth.ONT_RAPID, synth.ONT_RAPID_RC]
    adapters_note = "PacBio blunt + reverse complement" if hifi else "ONT rapid + reverse complement"
    if args.adapters == 4:
        # distinct 5' and 3' adapters with their reverse complements: what the automatic identification hands over for a
        # ligation kit (src/TGSFilter.cpp:3105-3113) -- four independent columns a pass in the middle scan
        # (the second pair: the first 50 bases of the ONT 1D^2 adapter and their reverse complement -- as long as the rapid
        # adapter, so that all four are searched within the same 16 differences at the default -M 35 and share one pass of
        # the middle scan; the whole 59-bp adapter at -M 35 is searched within 25 of 59: half of all random columns qualify,
        # a shape of the candidate lists, not of the column)
        wl_adapters = wl_adapters + [b"GGCGTCTGCTTGGGTGTTTAACCTTTTTGTCAGAGAGGTTCCAAGTCAGAG", synth.revcomp(b"GGCGTCTGCTTGGGTGTTTAACCTTTTTGTCAGAGAGGTTCCAAGTCAGAG")]
        adapters_note += " + the first 50 bases of ONT 1D^2 + reverse complement"
    # every rank builds the SAME two batches (same seeds): the fixed job is steps x these batches, dealt over ranks
    batches = [gen_batch(torch, device, args.reads, b + 1, mean_len, args.max_len, args.workload) for b in range(2)]
    max_bases = max(b["bases"] for b in batches)
    max_len = max(int(b["h_lens"].max()) for b in batches)
    p_kwargs = dict(adapters=wl_adapters, min_len=1000, min_q=20.0 if hifi else 10.0, head_trim=head_trim, tail_trim=tail_trim,
                    min_repeat=args.min_repeat, kmer=args.kmer)
    if args.short_adapters:
        # for information: a ligation-kit adapter pair of 28 bp (src/TGSFilter.cpp:2974-2975) with -M 24 -- adapters of at
        # most 32 bp take the one-dword column in the middle scan (the default -M 35 never searches the middle for them;
        # -M 24 allows 5 differences, about the error share of the default thresholds for the 50-bp adapters)
        p_kwargs.update(adapters=[b"AATGTACTTCGTTCAGTTACGTATTGCT", b"AGCAATACGTAACTGAACGAAGTACATT"], mid_match_len=24)
        flags += " -a ligation28.fa -M 24"
        adapters_note = "ONT ligation 28 bp + reverse complement"
        if args.adapters == 4:                       # ... and the 22-bp pair beside it (-M 20: both are searched in the middle)
            p_kwargs.update(adapters=p_kwargs["adapters"] + [b"GCAATACGTAACTGAACGAAGT", b"ACTTCGTTCAGTTACGTATTGC"], mid_match_len=22)
            flags = flags.replace("-M 24", "-M 22").replace("ligation28.fa", "ligation28+22.fa")
            adapters_note += " + ONT ligation 22 bp + reverse complement"
    p = abi.make_params(args.workload, max_batch_bases=max_bases + 64, max_batch_reads=args.reads, max_read_len=max_len, **p_kwargs)
    # what the PMC summaries of profiles/traffic.json are keyed by: the shape of a kernel-path step
    signature = "%s:reads=%d:mean=%d:p=%d:k=%d:trim=%d/%d%s%s" % (args.workload, args.reads, int(mean_len), args.min_repeat, args.kmer if args.min_repeat else 0,
                                                                 head_trim, tail_trim, ":short-adapters" if args.short_adapters else "", ":a4" if args.adapters == 4 else "")
    NS = max(1, args.streams)
    ctxs = [capi.Context(p, local_rank) for _ in range(NS)]
    comm = None
    state_rccl_ranks = None
    if world > 1:
        tdist.check_layout(ctxs[0].ctr_words, group=host_group)     # setup: same tally layout on every rank
        if args.backend == "nccl":
            # RCCL communicator of the job: rank 0 draws the id, the host-side group hands it round
            box = [rccl.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=host_group)
            comm = rccl.comm_init_rank(box[0], rank, world)
            state_rccl_ranks = rccl.comm_count(comm)
    fcap = max_bases // 1000 + args.reads + 16
    # every context gets a stream of its own (never torch's null stream: the library's streams are non-blocking,
    # nothing orders the null stream against them)
    streams = [torch.cuda.Stream(device=device) for _ in range(NS)]
    outs = [dict(reads=torch.empty(args.reads * 32, dtype=torch.uint8, device=device),
                 frags=torch.empty(fcap * 24, dtype=torch.uint8, device=device),
                 nfr=torch.zeros(4, dtype=torch.int32, device=device),
                 h_reads=torch.empty(args.reads * 32, dtype=torch.uint8).pin_memory(),
                 h_frags=torch.empty(fcap * 24, dtype=torch.uint8).pin_memory()) for _ in range(NS)]
    # fixed job: K steps in all; rank r takes steps r, r + world, ...
    K = args.kernel_steps if world == 1 else args.job_steps
    W = args.kernel_warmup
    my_steps = list(range(rank, K, world))
    state = {"NS": NS}

    def step(i, j):
        b = batches[i % 2]
        k = j % state["NS"]
        o = outs[k]
        with torch.cuda.stream(streams[k]):
            ctxs[k].submit_device(b["seq"].data_ptr(), b["qual"].data_ptr(), b["offsets"].data_ptr(),
                                  b["lengths"].data_ptr(), b["n"], b["n_bytes"], o["reads"].data_ptr(),
                                  o["frags"].data_ptr(), fcap, o["nfr"].data_ptr(), streams[k].cuda_stream)
            o["h_reads"].copy_(o["reads"], non_blocking=True)   # the per-read records go back to the host every step
        return b["bases"], b["n"]

    def sync_streams():
        for s in streams:
            s.synchronize()

    def barrier():
        if world > 1:
            dist.barrier(group=host_group)

    for i in range(W):
        step(i, i)
    sync_streams()
    for c in ctxs:
        c.wait()
        c.reset_counters()
        c.profile(True)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    bases = reads = 0
    for j, i in enumerate(my_steps):
        b, n = step(i, j)
        bases += b
        reads += n
    sync_streams()                                   # every submit stream is idle before the tallies are read
    for c in ctxs:
        c.wait()
    if world > 1 and comm is not None:
        # the job's only exchange, through the product's own entry point: the contexts of this rank become one vector
        # in HBM (tgsf_counters_merge), then ONE sum all-reduce of it over RCCL / xGMI (libtgsf_rccl)
        for c in ctxs[1:]:
            ctxs[0].merge_from(c)
        rccl.allreduce_counters(ctxs[0], comm, rank, world, check_layout=False)
        total_ctr = ctxs[0].counters()
    else:
        total_ctr = tdist.merge_counters([c.counters() for c in ctxs])
        if world > 1:                                # validation path (gloo): the same single sum through torch.distributed
            total_ctr = tdist.allreduce_counters(total_ctr, rank, world, device=xdev)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0

    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=host_group)
        dt = float(tt.item())
        bb = torch.tensor([bases, reads], dtype=torch.int64)
        dist.all_reduce(bb, op=dist.ReduceOp.SUM, group=host_group)
        bases_all, reads_all = int(bb[0].item()), int(bb[1].item())
    else:
        bases_all, reads_all = bases, reads

    # sanity: every read was classified exactly once (low-Q, or one of the 8 adapter classes) ...
    drop = total_ctr[:17]
    assert int(drop[0]) + int(drop[2:10].sum()) == reads_all, (drop, reads_all)

    def harvest():
        st, nbat = {}, 0
        for c in ctxs:
            st_c, nb_c = c.stage_times()
            nbat += nb_c
            for k, v in st_c.items():
                st[k] = st.get(k, 0.0) + v
        return {k: v / max(nbat, 1) for k, v in st.items() if v > 0}, nbat

    timed_stage_ms, _ = harvest()
    # the same kernels with the GPU to themselves: a few more steps on ONE stream, same HIP events
    for c in ctxs:
        c.profile(False)
    nprof = 4
    ctxs[0].profile(True)
    state["NS"] = 1
    for i in range(nprof):
        step(i, 0)
    sync_streams()
    ctxs[0].wait()
    excl_stage_ms, _ = harvest()
    state["NS"] = NS
    # ... and a random slice of batch 0 agrees with the oracle, record for record
    oracle_note = None
    if rank == 0 and not args.no_oracle_check and args.min_repeat == 0:
        with torch.cuda.stream(streams[0]):
            step(0, 0)
            outs[0]["h_frags"].copy_(outs[0]["frags"], non_blocking=True)
        streams[0].synchronize()
        ctxs[0].wait()
        rr = outs[0]["h_reads"].numpy().view(abi.READ_RESULT_DTYPE)
        ff = outs[0]["h_frags"].numpy().view(abi.FRAGMENT_DTYPE)
        t1 = time.perf_counter()
        n_chk, b_chk = oracle_slice_check(torch, batches[0], p_kwargs, args.workload, rr, ff)
        oracle_note = "%d random reads (%.1f Mbases) of batch 0: per-read records and fragments identical to oracle/ (%.1f s)" % (
            n_chk, b_chk / 1e6, time.perf_counter() - t1)

    # the dominant kernel of THIS configuration: the longest stage on the critical path with the GPU to itself (the two
    # stages of the auxiliary stream run beside the middle scan) -- the middle scan at C2 / C3, the repeat gate at C5
    aux = ("end_tables_raw", "end_windows")
    dom = max((k for k in excl_stage_ms if k not in aux), key=lambda k: excl_stage_ms[k])
    KERNELS = {"mid_scan": ("k_mid_flat<AT, Hot|Hot32[, filter]> (Myers infix scan of the read middles; + k_mid_recheck behind a filtering pass; k_mid_scanw / k_mid_scan_wide for adapters beyond 64 bp)", "valu"),
               "repeat_gate": ("k_repeat (k <= 11) / k_repeat_keys (k 12..31): GetKmerCount per kept fragment", "valu"),
               "stats_raw": ("k_stats<raw> (CalcAvgQuality on every read)", "hbm"),
               "stats_clean": ("k_stats<clean> (CalcAvgQuality on the kept fragments)", "hbm")}
    dom_kernel, dom_bound = KERNELS.get(dom, ("stage '%s'" % dom, "latency"))
    t_dom = excl_stage_ms[dom] / 1e3
    # every kernel of a batch (one batch in flight): the two stages that run on the auxiliary stream, beside the middle
    # scan, count in the SUM; the critical path leaves them out
    t_all = sum(excl_stage_ms.values()) / 1e3
    t_crit = sum(v for k, v in excl_stage_ms.items() if k not in aux) / 1e3
    steps_mine = max(len(my_steps), 1)
    alg_bytes = 2.0 * (bases / steps_mine) + 32.0 * (reads / steps_mine)    # 2 B/base + 32 B/read
    contract = alg_bytes / t_dom / 1e9 if t_dom > 0 else 0.0
    traffic, valu, stale, traffic_all, profiled_as = None, None, None, None, None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        tr = json.load(open(tj))
        stale = tr.get("kernel_source_hash") != kernel_source_hash()
        sigs = tr.get("signatures") or {}
        sig = sigs.get(signature)
        if sig is None:
            # the pre-pass of another file of the same configuration may resolve a trim one base away (C5: 79 / 80): the
            # profiled shape of the same workload stands in, and the line says which
            import re
            base = re.sub(r":trim=\d+/\d+", "", signature)
            for k2, v2 in sigs.items():
                if re.sub(r":trim=\d+/\d+", "", k2) == base:
                    sig, profiled_as = v2, k2
                    break
        if not stale and sig:
            stg = (sig.get("stages") or {}).get(dom) or {}
            traffic = stg.get("hbm_bytes_per_batch")
            traffic_all = sig.get("hbm_bytes_per_batch_all_kernels")
            vi, va = stg.get("valu_insts_per_batch"), sig.get("valu_insts_per_batch_all_kernels")
            if vi and va and t_dom > 0:
                valu = {"kernel_valu_insts_per_launch": vi, "kernel_issue_cycles_per_simd": vi * 4 / 1024,
                        "kernel_min_clock_ghz_if_valu_only": vi * 4 / 1024 / t_dom / 1e9,
                        "pipeline_valu_insts_per_batch": va, "source": sig.get("source")}
    roofline = {
        # the scan issues VALU instructions back to back for its whole duration: it is VALU-issue bound, and the HBM
        # fractions below are what that leaves (named, so that none of them is mistaken for another)
        "bound": dom_bound, "kernel": dom_kernel, "stage": dom,
        "achieved": contract, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": contract / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_all_kernels_per_batch": traffic_all, "stale_profile": stale, "traffic_profiled_signature": profiled_as or signature,
        "fractions_of_hbm_peak": {
            "dominant_kernel_contract": contract / HBM_PEAK_GBS,                      # (2 B/base + 32 B/read) / scan time
            "pipeline": (alg_bytes / t_all / 1e9 / HBM_PEAK_GBS) if t_all > 0 else 0.0,  # same bytes / sum of ALL kernel time
            "dominant_kernel_actual_hbm": (traffic / t_dom / 1e9 / HBM_PEAK_GBS) if traffic and t_dom > 0 else None,  # PMC bytes / scan time
            "whole_job": (2.0 * bases_all / world + 32.0 * reads_all / world) / dt / 1e9 / HBM_PEAK_GBS,
        },
        "valu_issue": valu,
        "measured": "HIP events on the launch stream around every stage (inside libtgsf); kernel durations of %d "
                    "single-stream steps run right after the timed region; 'timed_region_stage_ms' are the same events "
                    "inside the timed region with %d batches in flight; traffic/valu_issue come from the PMC passes in "
                    "profiles/ and are reported only while profiles/traffic.json carries the hash of today's kernel "
                    "sources (stale_profile says so)" % (nprof, NS),
        "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": excl_stage_ms[dom],
        "sum_kernel_ms": t_all * 1e3, "critical_path_ms": t_crit * 1e3,
        "stage_ms_per_step": excl_stage_ms, "timed_region_stage_ms": timed_stage_ms,
    }
    kp = {
        "what": "device-resident filter throughput: batches already in HBM, tgsf_submit_device, per-read records copied "
                "back every step; NOT the end-to-end metric",
        "value": bases_all / dt / 1e9, "unit": "Gbases/s", "steps": K, "warmup": W, "ms_per_step": dt / max(len(my_steps), 1) * 1e3,
        "scaling": "strong (fixed job of %d batches dealt over %d ranks)" % (K, world) if world > 1 else "single GPU",
        "workload": ("C3 shape: synthetic HiFi reads N(%.0f,/6) bp, %s, adapters %s" % (mean_len, flags, adapters_note)) if hifi else
                    ("C2: synthetic ONT reads, lognormal mean %.0f bp, %s, adapters %s" % (mean_len, flags, adapters_note)),
        "signature": signature, "reads_per_step": args.reads, "gbases_per_step": bases / steps_mine / 1e9, "batches_in_flight_per_gpu": NS,
        "oracle_check": oracle_note,
        "tally_exchange": ("one SUM all-reduce of the tally vector over RCCL (libtgsf_rccl), inside the timed region" if comm is not None else
                           ("one SUM all-reduce through torch.distributed/%s (validation path)" % args.backend if world > 1 else "none (one rank)")),
        "rccl_ranks": state_rccl_ranks,
        # the job's merged tallies: the same fixed job must give the same vector whatever the number of ranks
        "tallies": {"dropinfo": [int(x) for x in drop], "sha256_16": hashlib.sha256(np.ascontiguousarray(total_ctr).tobytes()).hexdigest()[:16],
                    "reads": reads_all, "bases": bases_all},
    }
    for c in ctxs:
        c.close()
    if comm is not None:
        rccl.comm_destroy(comm)
    return kp, roofline


def h2d_peak(torch, device, seconds=0.6):
    """Host-to-device rate of this GPU's link from pinned memory, three streams at once (what three feeder threads reach):
    the roof of an end-to-end run at one GPU, whose text crosses the link once."""
    n = 256 << 20
    try:
        host = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(3)]
        dev = [torch.empty(n, dtype=torch.uint8, device=device) for _ in range(3)]
        streams = [torch.cuda.Stream(device=device) for _ in range(3)]
        for k in range(3):
            with torch.cuda.stream(streams[k]):
                dev[k].copy_(host[k], non_blocking=True)
        torch.cuda.synchronize()
        t0, moved = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            for k in range(3):
                with torch.cuda.stream(streams[k]):
                    dev[k].copy_(host[k], non_blocking=True)
            torch.cuda.synchronize()
            moved += 3 * n
        dt = time.perf_counter() - t0
        return {"h2d_peak_gb_per_s": moved / dt / 1e9, "how": "three streams copying 256-MB pinned buffers to the device for %.1f s, measured in this run (after the end-to-end leg)" % dt}
    except Exception as e:                                    # (a measurement beside the result: never the reason a run fails)
        log("bench: link measurement failed: %r" % (e,))
        return None


LINE_LIMIT = 4096            # bytes of the ONE stdout line (round 5's grew to 21 kB and the driver could not parse it)


def _sig(x, digits=6):
    """Floats to 6 significant digits, recursively (the line is read by a machine: no digits nobody measures)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _clip(text, n):
    return text if text is None or len(text) <= n else text[:n - 3] + "..."


def compact_line(out, detail_path=None):
    """The one stdout line: the contract's keys, `roofline`, `cpu_baseline`, the speed-up and the scaling figures -- at most
    LINE_LIMIT bytes.  Everything else bench.py measures (per-run wall times, control-group records, per-sink blocks,
    tallies, the long descriptions) is in the detail file and on stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    c = {k: out.get(k) for k in keep}
    c["metric"] = _clip(c["metric"], 100)
    cfg = out.get("config") or {}
    c["config"] = {"workload": _clip(cfg.get("workload"), 300), "parallelism": _clip(cfg.get("parallelism"), 160)}
    r = out.get("roofline")
    if r:
        c["roofline"] = {k: r.get(k) for k in ("bound", "kernel", "stage", "achieved", "peak", "unit", "frac", "kernel_ms", "algorithmic_bytes_per_launch",
                                               "traffic", "traffic_all_kernels_per_batch", "fractions_of_hbm_peak", "sum_kernel_ms")}
        c["roofline"]["kernel"] = _clip(r.get("kernel"), 80)
    b = out.get("cpu_baseline")
    if b:
        c["cpu_baseline"] = {k: b.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
        c["cpu_baseline"]["sample"] = _clip(b.get("sample"), 200)
    sp = out.get("e2e_speedup_vs_reference")
    if sp:
        c["e2e_speedup_vs_reference"] = {"tmpfs_file": sp.get("tmpfs_file")}
    if out.get("scaling_figures"):
        c["scaling_figures"] = out["scaling_figures"]
    e2e = out.get("e2e") or {}
    if e2e.get("skipped"):
        c["skipped_legs"] = len(e2e["skipped"])
    c["bench_wall_s"] = out.get("bench_wall_s")
    if detail_path:
        c["detail"] = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT + os.sep) else detail_path
    c = _sig(c)
    line = json.dumps(c, separators=(",", ":"))
    for victim in ("detail", "scaling_figures", "e2e_speedup_vs_reference"):      # (never needed so far: a guard, not a plan)
        if len(line) <= LINE_LIMIT:
            break
        c.pop(victim, None)
        line = json.dumps(c, separators=(",", ":"))
    if len(line) > LINE_LIMIT:
        raise SystemExit("bench: the result line is %d bytes (limit %d)" % (len(line), LINE_LIMIT))
    return line


def emit(out, detail_path):
    """Writes the full record to the detail file (and to stderr, one line, prefixed) and returns the compact stdout line."""
    full = json.dumps(out)
    written = None
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                f.write(full + "\n")
            written = detail_path
        except OSError as e:
            log("bench: could not write %s: %r" % (detail_path, e))
    log("bench: detail: " + full)
    return compact_line(out, written)


def main():
    t_bench0 = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed end-to-end runs of the command line")
    ap.add_argument("--warmup", type=int, default=1, help="untimed end-to-end runs before them")
    ap.add_argument("--config", choices=sorted(E2E_CONFIGS), default="c2", help="BASELINE.json configuration of the end-to-end leg (c2: the headline; "
                    "c5: ultra-long reads, repeat gate and downsampling)")
    ap.add_argument("--e2e-reads", type=int, default=None, help="reads in the end-to-end FASTQ file (default: the configuration's -- C2: 4 M reads, ~90 KB "
                    "of text each); reduced -- and said so in config.workload -- to what the staging file system / the box's memory control group holds")
    ap.add_argument("--e2e-budget-s", type=float, default=900.0, help="seconds the end-to-end leg may take: optional legs are dropped (and named in "
                    "e2e.skipped) when it runs short; the K timed steps never are")
    ap.add_argument("--pinned-variant", action="store_true", help="also run the pinned-pre-pass variant (-5 0 -3 0 -a rapid.fa on 400 000 reads: round 2's headline)")
    ap.add_argument("--e2e-files", type=int, default=0, help="stage the end-to-end reads as this many consecutive files (default: as few as the box's memory allows)")
    ap.add_argument("--e2e-ranks", type=int, default=0, help="run the end-to-end command line as this many rank processes (tgsfilter --ranks; default: N for --gpus N > 1)")
    ap.add_argument("--no-sharded-leg", dest="sharded_leg", action="store_false", help="N = 1: skip the 3-ranks-on-one-GPU leg")
    ap.add_argument("--adapters", type=int, default=2, choices=[2, 4], help="kernel path: 2 = the configuration's adapter and its reverse complement; 4 = distinct 5' and 3' "
                    "adapters with their reverse complements (a ligation kit, src/TGSFilter.cpp:3105-3113), for information")
    ap.add_argument("--head-trim", type=int, default=None, help="kernel path: -5 (default: what the end-to-end pre-pass of this run resolved; 79 / 7 for the ONT / HiFi shape without one)")
    ap.add_argument("--tail-trim", type=int, default=None, help="kernel path: -3 (default: as --head-trim; 0 / 8)")
    ap.add_argument("--no-e2e", action="store_true", help="kernel path only (profiling runs); the headline is then the kernel path")
    ap.add_argument("--no-kernel-path", action="store_true")
    ap.add_argument("--kernel-steps", type=int, default=24)
    ap.add_argument("--kernel-warmup", type=int, default=3)
    ap.add_argument("--job-steps", type=int, default=31, help="N > 1: batches of the fixed C4 job (31 x 131072 = 4.06 M reads)")
    ap.add_argument("--reads", type=int, default=131072, help="reads per kernel-path step")
    ap.add_argument("--mean-len", type=float, default=None)
    ap.add_argument("--min-repeat", type=int, default=0, help="-p of config C5, for information")
    ap.add_argument("--short-adapters", action="store_true", help="kernel path with a 28-bp adapter pair and -M 24 (the one-dword scan column), for information")
    ap.add_argument("--kmer", type=int, default=11)
    ap.add_argument("--workload", choices=["ont", "hifi"], default="ont")
    ap.add_argument("--max-len", type=int, default=2_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-check", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the tally all-reduce (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="validation on a 1-GPU box: every rank uses device 0")
    ap.add_argument("--streams", type=int, default=3, help="kernel path: batches in flight per GPU")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"), help="where the full record goes (per-run wall times, per-sink blocks, stage "
                    "times, tallies ...): stdout carries ONE line of at most %d bytes" % LINE_LIMIT)
    args = ap.parse_args()
    if args.config == "c5":                # the kernel path in C5's shape: ultra-long reads through the repeat gate
        args.mean_len = args.mean_len or 150000.0
        args.min_repeat = args.min_repeat or 100
        if args.reads == 131072:
            args.reads = 32768             # (4.9 Gbases a step, as C2's steps)
    if args.config == "c3":                # the kernel path in C3's shape: HiFi reads, blunt adapter pair, -M 35 -T 50 (the defaults)
        args.workload = "hifi"
        if args.reads == 131072:
            args.reads = 262144            # (4.7 Gbases a step)
    # stdout carries exactly one line, the result: whatever libraries print there meanwhile goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0
    # (several processes on the GPUs of one node: this pool's host driver only supports dmabuf IPC -- RCCL's and HIP's
    # cross-process sharing need this; a value already set stays.  Before torch / any child process is started.)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    host_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # host-side group (barriers, timing) first: it does not touch the GPU
        # (the other ranks wait at a barrier while rank 0 runs the end-to-end leg: well beyond gloo's default 30 minutes)
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(hours=3))
        host_group = dist.group.WORLD

    # ---- end-to-end leg first (rank 0; subprocesses only, this process has not touched the GPU yet) ----
    e2e = None
    if not args.no_e2e:
        if rank == 0:
            if world > 1:
                args.no_cpu_baseline = True          # the reference is timed at N = 1 only (the contract: cpu_baseline on rank 0 at N=1)
            if args.share_gpu and not args.e2e_ranks:
                args.e2e_ranks = world               # (validation on a 1-GPU box: the N rank processes share device 0)
            e2e = e2e_leg(args, 1 if args.share_gpu else world)
        if world > 1:
            dist.barrier(group=host_group)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    xdev = device if args.backend == "nccl" else None

    kp = roofline = None
    if e2e and args.head_trim is None and args.tail_trim is None:
        # the kernel-path shape takes the trims the end-to-end pre-pass of this very run resolved
        import re
        for l in e2e["sinks"]["tmpfs_file"].get("info_prepass", []):
            m = re.match(r"INFO: trim ([53])' end length: (\d+)", l)
            if m:
                setattr(args, "head_trim" if m.group(1) == "5" else "tail_trim", int(m.group(2)))
    if world > 1:                                    # (every rank runs the same shape: rank 0's end-to-end leg decides)
        box = [args.head_trim, args.tail_trim]
        dist.broadcast_object_list(box, src=0, group=host_group)
        args.head_trim, args.tail_trim = box
    if not args.no_kernel_path:
        kp, roofline = kernel_leg(args, torch, dist, world, rank, local_rank, device, xdev, host_group)

    # the roof the end-to-end run sits under at one GPU: the link to the device (the text goes over it once)
    link = None
    if rank == 0 and e2e:
        link = h2d_peak(torch, device)

    if rank == 0:
        if e2e:
            s = e2e["sinks"]["tmpfs_file"]
            value, ms, steps, warmup = s["gbases_per_s"], s["wall_s_mean"] * 1e3, s["runs"], args.warmup
            metric = "filtered Gbases/sec (end-to-end, excl. gzip I/O)"
            cfg = E2E_CONFIGS[args.config]
            shape = "HiFi reads N(%.0f kb,/6)" % (cfg["mean_len"] / 1e3) if cfg.get("kind") == "hifi" else "ONT reads, lognormal mean %.0f kb" % (cfg["mean_len"] / 1e3)
            # (<= 300 characters: the long form -- why the reads are split, what was reduced -- is `workload_long` in the detail file)
            workload = ("%s end to end: %d of its %d synthetic %s as %d tmpfs file(s); bin/tgsfilter %s -t %d%s -> FASTQ on tmpfs + report; "
                        "step = one run over one file (%.0f Gbases), K steps dealt over the files%s"
                        % (cfg["name"], e2e["reads"], cfg["reads"], shape, e2e["files"], e2e["flags"], e2e["threads"],
                           " --ranks %d" % e2e["ranks"] if e2e.get("ranks") else "", s["bases_per_step_mean"] / 1e9, "; REDUCED to fit the box" if e2e.get("reduced") else ""))
            how = ("%d rank processes (tgsfilter --ranks %d, one per GPU: byte-range shards of each file, a part file per rank, pre-pass on rank 0, one "
                   "all-reduce of the tallies, rank 0 writes the report)" % (e2e["ranks"], e2e["ranks"])) if e2e.get("ranks") else \
                  "one process, every mapping taken down before it returns"
            e2e["workload_long"] = ("%s (%s; %s): %.2f Gbases, %.1f GB of FASTQ text%s%s; output %.1f GB" % (
                workload, cfg["what"], how, e2e["bases"] / 1e9, e2e["fastq_bytes"] / 1e9, ("; REDUCED: " + e2e["reduced"]) if e2e.get("reduced") else "",
                ("; " + e2e["split"]) if e2e.get("split") else "", s.get("output_bytes", 0) / 1e9))
        else:
            value, ms, steps, warmup = kp["value"], kp["ms_per_step"], kp["steps"], kp["warmup"]
            metric = "device-resident filter throughput (Gbases/sec, inputs in HBM; NOT end-to-end)"
            workload = kp["workload"] + "; %d reads (%.2f Gbases) per step, inputs resident in HBM" % (kp["reads_per_step"], kp["gbases_per_step"])
        out = {
            "metric": metric, "value": value, "unit": "Gbases/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload,
                       "parallelism": ("%d rank processes, one per GPU: byte-range shards of the input, part files, one all-reduce of the tallies" % world) if world > 1
                                      else "one GPU (three contexts / feeder threads); reads shard over rank processes with tgsfilter --ranks N"},
        }
        if e2e:
            out["e2e"] = e2e
            s = e2e["sinks"]["tmpfs_file"]
            if link:
                # how far each sink is from the roof that binds the end-to-end run: the text crosses the link once
                e2e["link"] = link
                for name, leg in e2e["sinks"].items():
                    text = leg["bases_per_step_mean"] / e2e["bases"] * e2e["fastq_bytes"] if name == "tmpfs_file" else s["per_file"][0]["fastq_bytes"]
                    leg["text_gb_per_s"] = text / leg["wall_s_mean"] / 1e9
                    leg["link_fraction"] = leg["text_gb_per_s"] / (link["h2d_peak_gb_per_s"] * (world if leg.get("ranks") and world > 1 else 1))
            if world > 1:
                e2e["scaling_note"] = ("every rank instantiates the pages of its own part file (N inodes, N fallocate streams) and feeds its own GPU over its own link; "
                                       "what the ranks share is the host: the page cache the text is read from, the memory bandwidth of the fill threads and the box's CPU "
                                       "quota (this box: %s CPUs for %d ranks)" % (e2e["box"]["cgroup_cpus"], world))
            if "reference_gbases_per_s" in s:
                rbt = s.get("reference_runs_by_threads_first_file") or {}
                out["cpu_baseline"] = {
                    "value": s["reference_gbases_per_s"], "unit": "Gbases/s", "cores": s["reference_threads"], "kind": "reference",
                    # (<= 200 characters; the long form is `sample_long`)
                    "sample": "oracle/_ref/tgsfilter_ref -t %d on the same %d file(s) (%.0f Gbases), same flags and sink, once each: %.0f s; records and INFO lines equal ours (asserted)"
                              % (s["reference_threads"], e2e["files"], e2e["bases"] / 1e9, s["reference_wall_s"]),
                    "runs_by_threads": rbt,
                    "sample_long": "the whole end-to-end workload (%d reads in %d file(s), %.2f Gbases, %.1f GB FASTQ on tmpfs), same flags, same sink (tmpfs file), "
                              "oracle/_ref/tgsfilter_ref: on the first file at %s, %d run(s) each; value = all bases / the reference's wall time over all files at "
                              "-t %d, the FASTER of them (one run per file): %.1f s; output multiset and INFO lines "
                              "(automatic trims, identified adapter, depths, counters) identical to ours for every file (asserted)"
                              % (e2e["reads"], e2e["files"], e2e["bases"] / 1e9, e2e["fastq_bytes"] / 1e9,
                                 " and ".join("-t %s (%s)" % (k2, v["why"]) for k2, v in rbt.items()), max([len(v["wall_s_runs"]) for v in rbt.values()] or [1]),
                                 s["reference_threads"], s["reference_wall_s"])}
                out["e2e_speedup_vs_reference"] = {
                    "tmpfs_file": s.get("speedup_vs_reference"),
                    "against": "the reference at -t %d: the faster of %s on this box" % (s["reference_threads"], ", ".join("-t " + k2 for k2 in rbt) or "its clamp"),
                    "by_reference_threads_first_file": {k2: v["wall_s_mean"] / s["per_file"][0]["wall_s_mean"] for k2, v in rbt.items()},
                    "dev_null_vs_reference_file_run_first_file": e2e["sinks"].get("dev_null", {}).get("speedup_vs_reference_file_run_first_file"),
                    "pinned_prepass_variant": e2e.get("variants", {}).get("pinned_prepass", {}).get("speedup_vs_reference")}
        if kp:
            out["kernel_path"] = kp
            out["roofline"] = roofline
            if world > 1 and args.backend == "nccl" and kp.get("rccl_ranks") != world:
                raise SystemExit("bench: the job's RCCL communicator has %s ranks, not %d" % (kp.get("rccl_ranks"), world))
        # the figures a scaling curve over N is read from, side by side: the end-to-end file sink (N > 1: one rank process and
        # one part file per GPU), the /dev/null sink, and the device-resident kernel path (the fixed job dealt over the
        # ranks, one tally all-reduce over RCCL)
        sharded = [v for k2, v in (e2e["sinks"].items() if e2e else []) if k2.startswith("tmpfs_part_files")]
        out["scaling_figures"] = {
            "n_gpus": world,
            "e2e_tmpfs_file_gbases_per_s": e2e["sinks"]["tmpfs_file"]["gbases_per_s"] if e2e else None,
            "e2e_ranks": e2e.get("ranks") if e2e else None,
            "e2e_dev_null_gbases_per_s": e2e["sinks"].get("dev_null", {}).get("gbases_per_s") if e2e else None,
            "e2e_3_ranks_sharing_one_gpu_part_files_gbases_per_s": sharded[0]["gbases_per_s"] if sharded else None,
            "kernel_path_gbases_per_s": kp["value"] if kp else None,
            "kernel_path_rccl_ranks": kp.get("rccl_ranks") if kp else None}
        # (what the whole invocation took, staging of the synthetic files and the reference's runs included: the driver's clock
        # around this process reads a little more -- interpreter start, the first import of torch)
        out["bench_wall_s"] = round(time.perf_counter() - t_bench0, 1)
        sys.stdout.flush()
        line = emit(out, args.detail_file)
        os.write(real_stdout, (line + "\n").encode())
    if world > 1:
        dist.barrier(group=host_group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
