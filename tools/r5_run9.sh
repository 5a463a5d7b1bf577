cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 1500 python tests/manual/live_campaign_sharded.py 20000 20250 gpu 300 > gpurun_out/r5i/campaign_sharded_gpu.txt 2>&1
tail -3 gpurun_out/r5i/campaign_sharded_gpu.txt
REPS=3 timeout 900 python tests/manual/e2e_shard.py 1333334 2 3 > gpurun_out/r5i/e2e_shard_1333k.txt 2>&1
head -8 gpurun_out/r5i/e2e_shard_1333k.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5i/pytest.log 2>&1
tail -4 gpurun_out/r5i/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5i/smoke.log 2>&1; tail -2 gpurun_out/r5i/smoke.log
