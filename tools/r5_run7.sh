cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5g/bench_driver_form.json 2> gpurun_out/r5g/bench_driver_form.err
tail -c 400 gpurun_out/r5g/bench_driver_form.err
timeout 1500 python -m pytest tests/test_cli_shard.py tests/test_gpu_fullsize.py -m gpu -q -k "config5 or end_to_end_leg" > gpurun_out/r5g/pytest_new.log 2>&1
tail -3 gpurun_out/r5g/pytest_new.log
