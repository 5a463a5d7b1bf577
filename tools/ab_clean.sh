#!/bin/bash
# A/B of the clean-table strategies on the kernel path (device-resident batches), the headline shapes WITH the trims the
# end-to-end pre-pass resolves (C2: -5 79 -3 0; C3: -5 7 -3 8), one batch in flight and three:
#   direct      TGSF_CLEAN_TABLES=direct     every kept fragment scanned a second time
#   byproduct   TGSF_CLEAN_TABLES=byproduct  every batch speculates: the raw pass tallies the expected fragment (round 6: split bins)
#   default     (unset)                      the device decides per batch whether the next one speculates
# Run on the GPU box: tools/ab_clean.sh [sittings] > gpurun_out/<tag>_clean_tables_ab.txt
export TGSF_DEBUG_KNOBS=1   # the test settings below are read only under this switch
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/ab_clean
rm -rf $O; mkdir -p $O
K="--no-e2e --no-cpu-baseline --no-oracle-check"
for sit in $(seq 1 ${1:-2}); do
for shape in "c2" "c3"; do
  for st in 1 3; do
    for mode in direct byproduct default; do
      if [ $mode = default ]; then unset TGSF_CLEAN_TABLES; else export TGSF_CLEAN_TABLES=$mode; fi
      python3 bench.py $K --config $shape --streams $st --detail-file $O/${shape}_s${st}_${mode}_$sit.json > /dev/null 2> $O/${shape}_s${st}_${mode}_$sit.err
    done
  done
done
done
unset TGSF_CLEAN_TABLES
python3 - <<'PY'
import glob, json, os
O = "gpurun_out/ab_clean"
print("%-34s %9s %8s | %9s %9s %9s %9s %9s | %7s %7s" % ("run", "Gbases/s", "ms/step", "stats_raw", "end_win", "mid_scan", "stats_cln", "sum(all)", "contract", "pipeline"))
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
        continue
    r = d["roofline"]; st = r["stage_ms_per_step"]
    print("%-34s %9.1f %8.3f | %9.3f %9.3f %9.3f %9.3f %9.3f | %7.3f %7.3f" % (os.path.basename(f)[:-5], d["value"], d["ms_per_step"], st.get("stats_raw", 0), st.get("end_windows", 0),
          st.get("mid_scan", 0), st.get("stats_clean", 0), r["sum_kernel_ms"], r["fractions_of_hbm_peak"]["dominant_kernel_contract"], r["fractions_of_hbm_peak"]["pipeline"]))
    if f.endswith("_1.json"):
        print("      ", d["kernel_path"]["workload"][:200])
PY
