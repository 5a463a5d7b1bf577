cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_cli_shard.py -m gpu -x -q > gpurun_out/r5b/pytest_shard.log 2>&1
tail -5 gpurun_out/r5b/pytest_shard.log
cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/memory.max > gpurun_out/r5b/box.txt 2>&1
nproc >> gpurun_out/r5b/box.txt
timeout 1500 python tests/manual/e2e_shard.py 700000 2 3 4 > gpurun_out/r5b/e2e_shard.txt 2>&1
tail -40 gpurun_out/r5b/e2e_shard.txt
