#!/bin/bash
# Sweep of the co-scheduling knobs: a high-priority side stream for the HBM-bound stats kernels (TGSF_STATS_PRIO),
# a lowest-priority side stream for the stats AND scan kernels so that the small latency-bound kernels of the
# other batches are dispatched first (TGSF_BIG_LOWPRIO), the stats grid, batches in flight.
out=${1:-gpurun_out/sweep_overlap.txt}
: > $out
for st in 2 3 4; do
for mode in none big_low stats_hi; do
  case $mode in none) e="";; big_low) e="TGSF_BIG_LOWPRIO=1";; stats_hi) e="TGSF_STATS_PRIO=1";; esac
  for rep in 1 2; do
  v=$(env $e python bench.py --steps 24 --warmup 4 --streams $st --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3))')
  echo "streams=$st mode=$mode -> $v" >> $out
  done
done; done
