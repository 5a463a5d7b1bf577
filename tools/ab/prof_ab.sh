#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the kernel path with the shipped library and with a variant
export TGSF_DEBUG_KNOBS=1 TGSF_CLEAN_TABLES=byproduct
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
K="--no-e2e --no-cpu-baseline --no-oracle-check --streams 1 --config ${2:-c2} --kernel-steps 8"
for lib in main variant; do
  if [ $lib = variant ]; then export TGSF_LIB=$R/tools/ab/$1; else unset TGSF_LIB; fi
  rm -rf /tmp/kt_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$lib -- python3 $R/bench.py $K --detail-file /tmp/kt_$lib.json > /dev/null 2> /tmp/kt_$lib.err
  echo "== $lib"
  python3 - /tmp/kt_$lib <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in [x for x in rows if "tgsf::" in x["Name"]][:16]:
    print("%-70s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
