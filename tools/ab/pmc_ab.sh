#!/bin/bash
# PMC counters of k_stats<true> (the clean pass) with the shipped library and with a variant
export TGSF_DEBUG_KNOBS=1 TGSF_CLEAN_TABLES=byproduct
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
K="--no-e2e --no-cpu-baseline --no-oracle-check --streams 1 --config c2 --kernel-steps 2 --kernel-warmup 1"
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCC_ATOMIC_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
for lib in main variant; do
  if [ $lib = variant ]; then export TGSF_LIB=$R/tools/ab/$1; else unset TGSF_LIB; fi
  rm -rf /tmp/pmc_$lib
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$lib -- python3 $R/bench.py $K --detail-file /tmp/p.json > /dev/null 2> /tmp/pmc_$lib.err
  python3 - /tmp/pmc_$lib $lib <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_stats<true" not in k and "k_stats<false" not in k: continue
    acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:40], r["Counter_Name"])] += 1
for k, d in acc.items():
    print(sys.argv[2], k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done; done
