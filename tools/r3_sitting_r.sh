cd $GRAFT_REPO_ROOT
timeout 900 python tests/manual/live_campaign.py 98000 98400 60 > gpurun_out/r3_campaign_c.txt 2>&1; tail -3 gpurun_out/r3_campaign_c.txt
TGSF_DOWN_EARLY_MIN=1 TGSF_DOWN_FEEDERS=2 TGSF_DOWN_BATCH_BYTES=1200000 TGSF_DOWN_MAP_MIN=1 TGSF_STRIDE_BYTES=50000 TGSF_BATCH_BYTES=40000 TGSF_FILL_MIN_BYTES=1 TGSF_POOL_CAP=3 timeout 600 python tests/manual/live_campaign.py 99000 99250 40 > gpurun_out/r3_campaign_d.txt 2>&1; tail -3 gpurun_out/r3_campaign_d.txt
timeout 600 python tests/manual/e2e_hifi.py 2000000 > gpurun_out/r3_e2e_hifi.txt 2>&1; grep -E "reads|wall|same" gpurun_out/r3_e2e_hifi.txt | cut -c1-200
