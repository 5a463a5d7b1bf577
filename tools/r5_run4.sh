cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5d/pytest.log 2>&1
tail -4 gpurun_out/r5d/pytest.log
bash tools/r5_ab_clean.sh > gpurun_out/r5d/ab_clean.txt 2>&1
cat gpurun_out/r5d/ab_clean.txt
