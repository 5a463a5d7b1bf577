#!/bin/bash
# After `tools/round_end_gpu.sh profile <tag>` came back through gpurun: copy the summaries that are to be judged from
# gpurun_out/ (scratch) into profiles/ (tracked) and merge the shapes' PMC traffic into profiles/traffic.json.
#   tools/collect_profiles.sh [tag]      (run in the repo root, in the container)
tag=${1:-r06}
cd "$(dirname "$0")/.."
for shape in "" _c3 _c5 _a4; do
  d=gpurun_out/prof_${tag}${shape}
  [ -d $d ] || { echo "missing $d"; exit 1; }
  cp $d/${tag}${shape}_kt_default_kernel_stats.csv $d/${tag}${shape}_kt_single_kernel_stats.csv $d/${tag}${shape}_pmc_hbm_traffic.csv $d/${tag}${shape}_pmc_valu.csv profiles/
  cp $d/kt_single_detail.json profiles/${tag}${shape}_single_stream_bench.json
  cp $d/kt_default_detail.json profiles/${tag}${shape}_three_in_flight_bench.json
done
python3 tools/merge_traffic.py gpurun_out/prof_${tag} gpurun_out/prof_${tag}_c3 gpurun_out/prof_${tag}_c5 gpurun_out/prof_${tag}_a4
for f in a4short a2short; do [ -s gpurun_out/${tag}_${f}_bench.json ] && cp gpurun_out/${tag}_${f}_bench.json profiles/; done
python3 -m pytest tests/test_profiles_fresh.py -q 2>&1 | tail -2
