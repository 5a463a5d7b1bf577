#!/bin/bash
export TGSF_DEBUG_KNOBS=1   # the test settings below are read only under this switch
out=gpurun_out/sweep_seg.txt; : > $out
for sc in 768 1024 1280 1536 2048; do
  for rep in 1 2; do
  v=$(TGSF_SEG_COLS=$sc python bench.py --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["stage_ms_per_step"]["mid_scan"],3))')
  echo "seg_cols=$sc -> $v" >> $out
  done
done
