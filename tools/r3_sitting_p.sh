cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/r3_gpu_suite3.txt 2>&1; tail -10 gpurun_out/r3_gpu_suite3.txt
TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.5 timeout 900 python tests/manual/fuzz_campaign.py 300000 301500 150 > gpurun_out/r3_fuzz_wide2.txt 2>&1; tail -3 gpurun_out/r3_fuzz_wide2.txt
TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.5 timeout 600 python tests/manual/fuzz_campaign.py 330000 331000 150 > gpurun_out/r3_fuzz_wide3.txt 2>&1; tail -3 gpurun_out/r3_fuzz_wide3.txt
bash tools/round_end_gpu.sh profile r03
