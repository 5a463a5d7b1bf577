#!/bin/bash
# The campaign sittings of round 3 (run through gpurun from the repo root; logs under gpurun_out/, the judged copies in profiles/):
#   fuzz: the library against the oracle over the widened domain; live: the command line against the reference binary.
export TGSF_DEBUG_KNOBS=1   # the test settings below are read only under this switch
cd $GRAFT_REPO_ROOT
TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.5 timeout 900 python tests/manual/fuzz_campaign.py 300000 301500 150 > gpurun_out/r3_fuzz_wide.txt 2>&1; tail -3 gpurun_out/r3_fuzz_wide.txt
TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.7 TGSF_FUZZ_MEAN_LEN=70000 timeout 600 python tests/manual/fuzz_campaign.py 310000 310250 30 > gpurun_out/r3_fuzz_long.txt 2>&1; tail -3 gpurun_out/r3_fuzz_long.txt
timeout 900 python tests/manual/live_campaign.py 98000 98400 60 > gpurun_out/r3_campaign_c.txt 2>&1; tail -3 gpurun_out/r3_campaign_c.txt
TGSF_DOWN_EARLY_MIN=1 TGSF_DOWN_FEEDERS=2 TGSF_DOWN_BATCH_BYTES=1200000 TGSF_DOWN_MAP_MIN=1 TGSF_STRIDE_BYTES=50000 TGSF_BATCH_BYTES=40000 TGSF_FILL_MIN_BYTES=1 TGSF_POOL_CAP=3 timeout 600 python tests/manual/live_campaign.py 99000 99250 40 > gpurun_out/r3_campaign_d.txt 2>&1; tail -3 gpurun_out/r3_campaign_d.txt
timeout 600 python tests/manual/e2e_hifi.py 2000000 > gpurun_out/r3_e2e_hifi.txt 2>&1; grep -E "reads|wall|same" gpurun_out/r3_e2e_hifi.txt | cut -c1-200
