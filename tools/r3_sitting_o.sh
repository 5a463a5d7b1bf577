cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r3_gpu_suite2.txt 2>&1; tail -14 gpurun_out/r3_gpu_suite2.txt
MEAN_LEN=150000 BASE_ARGS='-x ont -l 1000 -q 10 -g 3g -d 40 -p 100 -k 11' SHOW=DOWN,TIMING python tests/manual/e2e_threads.py 200000 2 '' 'TGSF_DOWN_FEEDERS=1' 'TGSF_NO_EARLY_RESERVE=1' 'TGSF_DETACH=1' > gpurun_out/r3_c5_down2.txt 2>&1; cat gpurun_out/r3_c5_down2.txt
bash tools/round_end_gpu.sh profile r03
