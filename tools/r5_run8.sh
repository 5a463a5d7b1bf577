cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
( time python bench.py --config c3 --steps 3 --warmup 1 ) > gpurun_out/r5h/bench_c3.json 2> gpurun_out/r5h/bench_c3.err
tail -c 300 gpurun_out/r5h/bench_c3.err
( time python bench.py --config c5 --steps 2 --warmup 1 ) > gpurun_out/r5h/bench_c5.json 2> gpurun_out/r5h/bench_c5.err
tail -c 300 gpurun_out/r5h/bench_c5.err
