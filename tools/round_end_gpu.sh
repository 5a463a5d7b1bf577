cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/gpu_suite.txt 2>&1; tail -16 gpurun_out/gpu_suite.txt
bash tools/profile_round.sh r02 > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gate11 -- python3 $R/bench.py --no-e2e --no-cpu-baseline --no-oracle-check --min-repeat 100 --kmer 11 --streams 1 --kernel-steps 6 > $R/gpurun_out/gate11_single.json 2> $R/gpurun_out/gate11.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gate21 -- python3 $R/bench.py --no-e2e --no-cpu-baseline --no-oracle-check --min-repeat 100 --kmer 21 --streams 1 --kernel-steps 6 > $R/gpurun_out/gate21_single.json 2> $R/gpurun_out/gate21.err
cd $R
find gpurun_out/prof_gate11 gpurun_out/prof_gate21 -name "*kernel_stats.csv" | head
for d in gate11 gate21; do f=$(find gpurun_out/prof_$d -name "*kernel_stats.csv" | head -1); head -8 "$f" > gpurun_out/${d}_kernel_stats_head.csv; done
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json
