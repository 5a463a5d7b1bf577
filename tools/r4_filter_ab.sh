#!/bin/bash
# round 4: A/B of the middle scan's 32-row filter (TGSF_MID_FILTER = 0 off, 1 / 2 test stride), kernel path only, on the
# HiFi shape (--config c3: the two 45-bp PacBio adapters, k = 11) and on the default ONT shape (k = 16: filter not used).
#   gpurun -- 'bash tools/r4_filter_ab.sh [tests]'   -> gpurun_out/r4_filter_ab.txt
export TGSF_DEBUG_KNOBS=1   # the test settings below are read only under this switch
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4_filter_ab.txt; : > $out
brief() { python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
    s=r['stage_ms_per_step']
    print('value %.1f Gbases/s  ms/step %.3f  mid_scan %.3f  sum_kernel_ms %.3f  stage %s frac %.4f' % (j['value'], j['ms_per_step'], s['mid_scan'], r['sum_kernel_ms'], r.get('stage'), r['frac']))
except Exception as e: print('failed', e)"; }
if [ "$1" = tests ]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "mid_filter or mid_scan_variants or golden" > gpurun_out/r4_filter_parity.txt 2>&1; echo "parity rc=$?" >> $out; tail -3 gpurun_out/r4_filter_parity.txt >> $out
  shift
fi
run() { cfg=$1; st=$2; shift 2; echo "== $cfg streams=$st $*" >> $out; env "$@" python bench.py --no-e2e --no-cpu-baseline --no-oracle-check --kernel-steps 12 --streams $st $cfg 2>gpurun_out/r4_fab.err | brief >> $out; }
for f in 0 1 2; do run "--config c3" 1 TGSF_MID_FILTER=$f; done
for f in 0 2 1; do run "--config c3" 3 TGSF_MID_FILTER=$f; done
run "" 1 TGSF_MID_FILTER=2
run "" 3 TGSF_MID_FILTER=2
