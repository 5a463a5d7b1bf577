#!/bin/bash
# The GPU sittings behind a round's profiles/ (run through gpurun from the repo root; outputs under gpurun_out/, the
# summaries to be judged are then copied into profiles/ by hand -- profiles/README.md says which):
#   tools/round_end_gpu.sh suite           the GPU test suite
#   tools/round_end_gpu.sh profile <tag>   kernel trace + PMC passes of the four kernel-path shapes (tools/profile_round.sh:
#                                          C2 and C3 with the pre-pass's trims, C5, 4 adapters), their traffic.json merged,
#                                          the short-adapter shapes
#   tools/round_end_gpu.sh ab <tag>        the clean-table A/B (tools/ab_clean.sh)
#   tools/round_end_gpu.sh bench           the full bench line as the driver runs it + --config c3 + --config c5
#   tools/round_end_gpu.sh shard           rank processes sharing the one GPU against the single process (tests/manual/e2e_shard.py)
cd $GRAFT_REPO_ROOT
what=${1:-suite}; tag=${2:-r06}
R=$GRAFT_REPO_ROOT
case $what in
suite)
  python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/gpu_suite.txt 2>&1; tail -4 gpurun_out/gpu_suite.txt ;;
profile)
  bash tools/profile_round.sh $tag > gpurun_out/profile_$tag.log 2>&1
  bash tools/profile_round.sh ${tag}_c3 --config c3 > gpurun_out/profile_${tag}_c3.log 2>&1
  bash tools/profile_round.sh ${tag}_c5 --config c5 > gpurun_out/profile_${tag}_c5.log 2>&1
  bash tools/profile_round.sh ${tag}_a4 --adapters 4 > gpurun_out/profile_${tag}_a4.log 2>&1
  cd $R
  python3 tools/merge_traffic.py gpurun_out/prof_$tag gpurun_out/prof_${tag}_c3 gpurun_out/prof_${tag}_c5 gpurun_out/prof_${tag}_a4
  python3 bench.py --no-e2e --no-cpu-baseline --adapters 4 --short-adapters --streams 1 --detail-file gpurun_out/${tag}_a4short_bench.json > gpurun_out/a4short.line 2> gpurun_out/a4short.err
  python3 bench.py --no-e2e --no-cpu-baseline --short-adapters --streams 1 --detail-file gpurun_out/${tag}_a2short_bench.json > gpurun_out/a2short.line 2> gpurun_out/a2short.err
  tail -c 300 gpurun_out/a4short.line ;;
ab)
  bash tools/ab_clean.sh > gpurun_out/${tag}_clean_tables_ab.txt 2>&1; head -30 gpurun_out/${tag}_clean_tables_ab.txt ;;
bench)
  python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file gpurun_out/${tag}_bench_detail.json > gpurun_out/${tag}_bench_line.json 2> gpurun_out/bench.err; tail -c 600 gpurun_out/${tag}_bench_line.json
  python bench.py --config c3 --steps 3 --warmup 1 --detail-file gpurun_out/${tag}_bench_c3_detail.json > gpurun_out/${tag}_bench_c3_line.json 2> gpurun_out/bench_c3.err; tail -c 400 gpurun_out/${tag}_bench_c3_line.json
  python bench.py --config c5 --steps 2 --warmup 1 --detail-file gpurun_out/${tag}_bench_c5_detail.json > gpurun_out/${tag}_bench_c5_line.json 2> gpurun_out/bench_c5.err; tail -c 400 gpurun_out/${tag}_bench_c5_line.json ;;
shard)
  REPS=3 python tests/manual/e2e_shard.py 1333334 2 3 > gpurun_out/${tag}_shard_one_gpu.txt 2>&1; head -12 gpurun_out/${tag}_shard_one_gpu.txt ;;
esac
