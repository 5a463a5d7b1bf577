#!/bin/bash
export TGSF_DEBUG_KNOBS=1 TGSF_CLEAN_TABLES=byproduct TGSF_TRACE_BP=1
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
K="--no-e2e --no-cpu-baseline --no-oracle-check --streams 1 --config c2 --kernel-steps 4"
for lib in main variant; do
  if [ $lib = variant ]; then export TGSF_LIB=$R/tools/ab/$1; else unset TGSF_LIB; fi
  echo "== $lib"; python3 bench.py $K --detail-file /tmp/t.json 2>&1 >/dev/null | grep "tgsf: clean tables"
done
