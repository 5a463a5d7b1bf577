#!/usr/bin/env python3
"""Thread / stride settings of the command line on a C2-shaped file on the GPU box, with the control group's CPU
accounting beside each run (cpu.max caps the box at 16 CPUs' worth of time: bursts beyond it stall every thread).
    python tests/manual/e2e_threads.py [reads=400000] [runs=3] 'ENV=V ENV2=V' ...      (ARGS="-t 16" adds arguments)"""
import os
import sys
import tempfile
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tgsfilter_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
print("box:", bench.cgroup_limits(), flush=True)
DEFAULT_CLI = bench.CLI
td = tempfile.mkdtemp(prefix="thr_", dir="/dev/shm")
try:
    fq = os.path.join(td, "in.fq")
    bases, nbytes = synth.write_ont_fastq(fq, n, seed=2, procs=32, mean_len=float(os.environ.get("MEAN_LEN", "45000")))
    base_args = os.environ.get("BASE_ARGS", "-x ont -l 1000 -q 10").split()
    show = os.environ.get("SHOW", "")            # e.g. SHOW=DOWN,TIMING prints those lines of the last run
    for setting in (sys.argv[3:] or [""]):
        env = dict(os.environ, TGSF_TIMING="1")
        bench.CLI = DEFAULT_CLI
        out = os.path.join(td, "out.fq")
        extra = []
        for kv in setting.split():
            k, v = kv.split("=", 1)
            if k == "ARGS":
                extra = v.replace(",", " ").split()
            elif k == "OUT":
                out = v
            elif k == "CLI":                      # another build of the command line (A/B in one sitting)
                bench.CLI = v if os.path.isabs(v) else os.path.join(ROOT, v)
            else:
                env[k] = v
        walls, cpus, thr, tim = [], [], [], {}
        for rep in range(runs):
            if os.path.isfile(out) and not os.path.islink(out):
                os.remove(out)
            dt, err = bench.run_cmd([bench.CLI, "-i", fq, "-o", out, "-t", "32"] + base_args + extra, env)
            walls.append(dt)
            cpus.append(bench.LAST_RUN_CPU.get("cpu_s", 0))
            thr.append(bench.LAST_RUN_CPU.get("throttled_periods", 0))
            tim = bench.parse_timing(err)
        print("[%-70s] wall min %.3f mean %.3f s (%.2f Gbases/s) | cpu %.1f s | throttled periods %s | fallocate %.2f populate %.2f index+prepass %.2f pipeline %.2f" % (
            setting, min(walls), sum(walls) / len(walls), bases / (sum(walls) / len(walls)) / 1e9, sum(cpus) / len(cpus), thr,
            tim.get("fallocate_s", 0), tim.get("populate_s", 0), tim.get("index_prepass_s", 0), tim.get("pipeline_s", 0)), flush=True)
        for l in err.splitlines():
            if show and l.split(":")[0] in show.split(","):
                print("    " + l[:1800], flush=True)
finally:
    shutil.rmtree(td, ignore_errors=True)
