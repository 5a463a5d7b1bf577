#!/bin/bash
# Sweep of the co-scheduling knobs (priority stream for the HBM-bound stats kernels, their grid,
# batches in flight).  Prints Gbases/s per combination.
out=${1:-gpurun_out/sweep_overlap.txt}
: > $out
for st in 2 3; do
for pr in 0 1; do
for sg in 768 512 256; do
  v=$(TGSF_STATS_PRIO=$pr TGSF_STATS_GRID=$sg python bench.py --steps 16 --warmup 3 --streams $st --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3), {k:round(v,2) for k,v in d["roofline"]["timed_region_stage_ms"].items() if k in ("stats_raw","mid_scan","stats_clean")})')
  echo "streams=$st prio=$pr stats_grid=$sg -> $v" >> $out
done; done; done
