#!/usr/bin/env python3
"""bench.py's end-to-end leg at a chosen size with a few runs, plus what the box offers (tmpfs room, huge-page
settings, cores): the sitting behind profiles/r03_host_e2e_c2_full*.txt.
    python tests/manual/e2e_c2_full.py [reads=4000000] [steps=2] [warmup=1]"""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for cmd in ("uname -r", "nproc", "df -h /dev/shm /tmp", "free -g", "cat /sys/kernel/mm/transparent_hugepage/enabled",
            "cat /sys/kernel/mm/transparent_hugepage/shmem_enabled", "cat /sys/kernel/mm/transparent_hugepage/defrag",
            "numactl -H", "cat /proc/sys/vm/overcommit_memory", "ulimit -l", "ulimit -n"):
    p = subprocess.run(cmd, shell=True, capture_output=True)
    print("$ %s\n%s" % (cmd, (p.stdout + p.stderr).decode().strip()), flush=True)
args = types.SimpleNamespace(e2e_reads=reads, steps=steps, warmup=warmup, no_cpu_baseline=os.environ.get("NO_REF") == "1",
                             e2e_budget_s=float(os.environ.get("BUDGET_S", "3000")),
                             no_pinned_variant=os.environ.get("NO_PINNED") == "1")
r = bench.e2e_leg(args, 1)
print(json.dumps(r, indent=1))
