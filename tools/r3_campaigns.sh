cd $GRAFT_REPO_ROOT
TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.5 timeout 800 python tests/manual/fuzz_campaign.py 300000 301500 150 > gpurun_out/r3_fuzz_wide.txt 2>&1; tail -3 gpurun_out/r3_fuzz_wide.txt
TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.7 TGSF_FUZZ_MEAN_LEN=70000 timeout 600 python tests/manual/fuzz_campaign.py 310000 310250 30 > gpurun_out/r3_fuzz_long.txt 2>&1; tail -3 gpurun_out/r3_fuzz_long.txt
timeout 900 python tests/manual/live_campaign.py 96000 96400 60 > gpurun_out/r3_campaign_a.txt 2>&1; tail -3 gpurun_out/r3_campaign_a.txt
TGSF_DETACH=1 TGSF_STREAM_MIN_BYTES=1 TGSF_CHUNK_BYTES=30000 TGSF_BATCH_BYTES=40000 TGSF_FILL_MIN_BYTES=1 TGSF_STRIDE_BYTES=60000 TGSF_POOL_CAP=3 timeout 600 python tests/manual/live_campaign.py 97000 97200 40 > gpurun_out/r3_campaign_b.txt 2>&1; tail -3 gpurun_out/r3_campaign_b.txt
