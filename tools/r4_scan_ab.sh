#!/bin/bash
# round 4: A/B of the middle scan (k_mid_scan1 against k_mid_flat and its knobs), single stream, kernel path only.
#   gpurun -- 'bash tools/r4_scan_ab.sh [tests]'   -> gpurun_out/r4_scan_ab.txt
export TGSF_DEBUG_KNOBS=1   # the test settings below are read only under this switch
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4_scan_ab.txt; : > $out
brief() { python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
    s=r['stage_ms_per_step']
    print('value %.1f Gbases/s  ms/step %.3f  mid_scan %.3f  sum_kernel_ms %.3f  frac %.4f' % (j['value'], j['ms_per_step'], s['mid_scan'], r['sum_kernel_ms'], r['frac']))
except Exception as e: print('failed', e)"; }
if [ "$1" = tests ]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r4_parity.txt 2>&1; echo "parity rc=$?" >> $out; tail -3 gpurun_out/r4_parity.txt >> $out
  shift
fi
run() { echo "== $*" >> $out; env "$@" python bench.py --no-e2e --no-cpu-baseline --no-oracle-check --kernel-steps 12 --streams 1 2>gpurun_out/r4_ab.err | brief >> $out; }
if [ $# -gt 0 ]; then
  for v in "$@"; do run $v; done
else
  run TGSF_MID_FLAT=0
  run TGSF_MID_FLAT=1
  run TGSF_MID_FLAT=1 TGSF_FLAT_SHARE8=4
  run TGSF_MID_FLAT=1 TGSF_FLAT_SHARE8=6
  run TGSF_MID_FLAT=1 TGSF_FLAT_PMIN=64
  run TGSF_MID_FLAT=1 TGSF_FLAT_PMIN=16
  run TGSF_MID_FLAT=1 TGSF_FLAT_PMAX=128
  run TGSF_MID_FLAT=1 TGSF_FLAT_BLOCKS=4
  run TGSF_MID_FLAT=1 TGSF_FLAT_BLOCKS=6
  run TGSF_MID_FLAT=0
  run TGSF_MID_FLAT=1
fi
cat $out
