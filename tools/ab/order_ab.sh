#!/bin/bash
export TGSF_DEBUG_KNOBS=1 TGSF_CLEAN_TABLES=byproduct
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
K="--no-e2e --no-cpu-baseline --no-oracle-check --streams 1 --config c2 --kernel-steps 6"
for lib in libtgsf_nosplit.so main libtgsf_maincopy.so main libtgsf_nosplit.so libtgsf_maincopy.so; do
  if [ $lib = main ]; then unset TGSF_LIB; else export TGSF_LIB=$R/tools/ab/$lib; fi
  python3 bench.py $K --detail-file /tmp/t.json > /dev/null 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python3 - $lib <<'PY'
import json,sys
d=json.load(open("/tmp/t.json")); st=d["roofline"]["stage_ms_per_step"]
print("%-22s %7.1f Gbases/s | raw %.3f clean %.3f scan %.3f sum %.3f" % (sys.argv[1], d["value"], st["stats_raw"], st["stats_clean"], st["mid_scan"], d["roofline"]["sum_kernel_ms"]))
PY
done
