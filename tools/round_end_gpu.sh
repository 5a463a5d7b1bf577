#!/bin/bash
# One GPU sitting behind a round's profiles/: the GPU suite, tools/profile_round.sh, gated profiles, the final bench line,
# the repeat-gate probe and the manual end-to-end scripts.  Run through gpurun from the repo root; outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/gpu_suite.txt 2>&1; tail -4 gpurun_out/gpu_suite.txt
bash tools/profile_round.sh r02 > gpurun_out/profile_round.log 2>&1; tail -2 gpurun_out/profile_round.log
cp gpurun_out/prof_r02/traffic.json profiles/traffic.json      # (on the box: the bench line below quotes the traffic of THIS build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in 11 21; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gate$k -- python3 $R/bench.py --no-e2e --no-cpu-baseline --no-oracle-check --min-repeat 100 --kmer $k --streams 1 --kernel-steps 6 > $R/gpurun_out/gate${k}_single.json 2> $R/gpurun_out/gate$k.err
done
cd $R
for k in 11 12 13 15 16 31; do python bench.py --no-e2e --no-cpu-baseline --min-repeat 100 --kmer $k --kernel-steps 6 > gpurun_out/b_repeat_k$k.json 2> gpurun_out/b_repeat.err; done
python bench.py --no-e2e --no-cpu-baseline --kernel-steps 6 > gpurun_out/b_nogate.json 2>> gpurun_out/b_repeat.err
timeout 1500 python tests/manual/e2e_c5.py 100000 > gpurun_out/e2e_c5.txt 2>&1; grep -E "flags|wall|same" gpurun_out/e2e_c5.txt | cut -c1-120
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 300 gpurun_out/bench_final.json
