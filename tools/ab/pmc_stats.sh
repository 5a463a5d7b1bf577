#!/bin/bash
# PMC counters of the raw pass (k_stats<raw ...>) with the by-product forced and without it, on one shape (default: c3)
export TGSF_DEBUG_KNOBS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
K="--no-e2e --no-cpu-baseline --no-oracle-check --streams 1 --config ${1:-c3} --kernel-steps 2 --kernel-warmup 1"
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE"; do
for mode in direct byproduct; do
  export TGSF_CLEAN_TABLES=$mode
  rm -rf /tmp/pmc_$mode
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$mode -- python3 $R/bench.py $K --detail-file /tmp/p.json > /dev/null 2> /tmp/pmc_$mode.err
  python3 - /tmp/pmc_$mode $mode <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_stats<false" not in k: continue
    acc[k[:44]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:44], r["Counter_Name"])] += 1
for k, d in acc.items():
    print(sys.argv[2], k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done; done
