cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5f/pytest.log 2>&1
tail -4 gpurun_out/r5f/pytest.log
bash tools/profile_round.sh r05 > gpurun_out/r5f/prof_r05.log 2>&1
bash tools/profile_round.sh r05_c3 --config c3 > gpurun_out/r5f/prof_r05_c3.log 2>&1
bash tools/profile_round.sh r05_c5 --config c5 > gpurun_out/r5f/prof_r05_c5.log 2>&1
bash tools/profile_round.sh r05_a4 --adapters 4 > gpurun_out/r5f/prof_r05_a4.log 2>&1
cd $GRAFT_REPO_ROOT
python3 bench.py --no-e2e --no-cpu-baseline --adapters 4 --short-adapters --streams 1 > gpurun_out/r5f/a4short_s1.json 2> gpurun_out/r5f/a4short_s1.err
python3 bench.py --no-e2e --no-cpu-baseline --short-adapters --streams 1 > gpurun_out/r5f/a2short_s1.json 2> gpurun_out/r5f/a2short_s1.err
ls gpurun_out/prof_r05*
