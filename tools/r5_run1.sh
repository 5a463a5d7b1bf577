cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
python bench.py --no-e2e --no-cpu-baseline --streams 1 > gpurun_out/r5a/kp_single.json 2> gpurun_out/r5a/kp_single.err
python bench.py --no-e2e --no-cpu-baseline > gpurun_out/r5a/kp_default.json 2> gpurun_out/r5a/kp_default.err
python bench.py --no-e2e --no-cpu-baseline --workload hifi --streams 1 > gpurun_out/r5a/kp_hifi_single.json 2> gpurun_out/r5a/kp_hifi_single.err
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5a/pytest.log 2>&1
tail -5 gpurun_out/r5a/pytest.log
