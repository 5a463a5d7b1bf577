#!/bin/bash
# A/B of the clean-table strategies on the kernel path (device-resident batches), the headline shapes WITH the trims the
# end-to-end pre-pass resolves (C2: -5 79 -3 0; C3: -5 7 -3 8), one batch in flight and three:
#   direct      TGSF_CLEAN_TABLES=direct  (round 4's way for trimmed reads: every kept fragment scanned a second time)
#   by-product  default                   (round 5: the raw pass tallies the expected fragment; corrections only)
# and the 4-adapter shapes.  Run on the GPU box: tools/ab_clean.sh > gpurun_out/r5_ab_clean.txt
export TGSF_DEBUG_KNOBS=1   # the test settings below are read only under this switch
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r5_ab
mkdir -p $O
K="--no-e2e --no-cpu-baseline --no-oracle-check"
for sit in 1 2; do
for shape in "c2" "c3"; do
  for st in 1 3; do
    for mode in direct byproduct; do
      if [ $mode = direct ]; then export TGSF_CLEAN_TABLES=direct; else unset TGSF_CLEAN_TABLES; fi
      python3 bench.py $K --config $shape --streams $st > $O/${shape}_s${st}_${mode}_$sit.json 2> $O/${shape}_s${st}_${mode}_$sit.err
    done
  done
done
done
unset TGSF_CLEAN_TABLES
python3 bench.py --no-e2e --no-cpu-baseline --adapters 4 --streams 1 > $O/a4_s1.json 2> $O/a4_s1.err
python3 bench.py --no-e2e --no-cpu-baseline --adapters 4 --streams 3 > $O/a4_s3.json 2> $O/a4_s3.err
python3 bench.py --no-e2e --no-cpu-baseline --adapters 4 --short-adapters --streams 1 > $O/a4short_s1.json 2> $O/a4short_s1.err
python3 bench.py --no-e2e --no-cpu-baseline --short-adapters --streams 1 > $O/a2short_s1.json 2> $O/a2short_s1.err
python3 bench.py --no-e2e --no-cpu-baseline --streams 1 > $O/a2_s1.json 2> $O/a2_s1.err
python3 - <<'PY'
import glob, json, os
O = "gpurun_out/r5_ab"
print("%-34s %9s %8s | %9s %9s %9s %9s %9s | %7s %7s" % ("run", "Gbases/s", "ms/step", "stats_raw", "end_win", "mid_scan", "stats_cln", "sum(all)", "contract", "pipeline"))
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
        continue
    r = d["roofline"]; st = r["stage_ms_per_step"]
    print("%-34s %9.1f %8.3f | %9.3f %9.3f %9.3f %9.3f %9.3f | %7.3f %7.3f" % (os.path.basename(f)[:-5], d["value"], d["ms_per_step"], st.get("stats_raw", 0), st.get("end_windows", 0),
          st.get("mid_scan", 0), st.get("stats_clean", 0), r["sum_kernel_ms"], r["fractions_of_hbm_peak"]["dominant_kernel_contract"], r["fractions_of_hbm_peak"]["pipeline"]))
    if f.endswith("_1.json") or "a4" in f or "a2" in f:
        print("      ", d["kernel_path"]["workload"][:200])
PY
