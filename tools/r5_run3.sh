cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_cli_shard.py -m gpu -q > gpurun_out/r5c/pytest_shard.log 2>&1
tail -3 gpurun_out/r5c/pytest_shard.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5c/bench_driver_form.json 2> gpurun_out/r5c/bench_driver_form.err
tail -c 600 gpurun_out/r5c/bench_driver_form.err
