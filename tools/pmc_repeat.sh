#!/bin/bash
# PMC passes for the repeat-gate kernels (VALU issue and LDS activity), one counter set per run, --kernel-trace only.
#   tools/pmc_repeat.sh   -> gpurun_out/pmc_repeat/*.csv, gpurun_out/pmc_repeat_summary.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_repeat
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for k in 11 21; do
  B="python3 $R/bench.py --no-e2e --no-cpu-baseline --no-oracle-check --min-repeat 100 --kmer $k --kernel-steps 2 --kernel-warmup 1 --streams 1"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $O/sq_k$k -- $B > $O/sq_k$k.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/lds_k$k -- $B > $O/lds_k$k.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for run in ("sq_k11", "lds_k11", "sq_k21", "lds_k21"):
    for fn in glob.glob(O + "/" + run + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_repeat" in r["Kernel_Name"]:
                rows[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O + "/../pmc_repeat_summary.csv", "w") as o:
    names = sorted({c for d in rows.values() for c in d})
    o.write("kernel,launches," + ",".join(names) + "\n")
    for k, d in rows.items():
        n = max(len(v) for v in d.values())
        o.write(k + "," + str(n) + "," + ",".join("%.0f" % (sum(d[c]) / len(d[c])) if d.get(c) else "" for c in names) + "\n")
print(open(O + "/../pmc_repeat_summary.csv").read())
PY
