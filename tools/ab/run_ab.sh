#!/bin/bash
# same box, alternating: the shipped library against a variant (TGSF_LIB), kernel path, by-product forced
export TGSF_DEBUG_KNOBS=1 TGSF_CLEAN_TABLES=byproduct
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
K="--no-e2e --no-cpu-baseline --no-oracle-check --streams 1"
for rep in 1 2; do for shape in c2 c3; do for lib in main variant; do
  if [ $lib = variant ]; then export TGSF_LIB=$R/tools/ab/$1; else unset TGSF_LIB; fi
  python3 bench.py $K --config $shape --detail-file /tmp/ab_${shape}_${lib}_$rep.json > /dev/null 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python3 - /tmp/ab_${shape}_${lib}_$rep.json $shape $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); st=d["roofline"]["stage_ms_per_step"]
print("%s %-8s %7.1f Gbases/s | raw %.3f clean %.3f scan %.3f prepare %.3f sum %.3f | %s" % (sys.argv[2], sys.argv[3], d["value"], st["stats_raw"], st["stats_clean"], st["mid_scan"], st["prepare+sort"], d["roofline"]["sum_kernel_ms"], d["kernel_path"]["tallies"]["sha256_16"]))
PY
done; done; done
