#!/bin/bash
# Campaign sittings on the build with the filtering middle scan (gpurun from the repo root; judged copies in profiles/): the library
# against the oracle (tests/fuzz.py: random parameter sets over the reference's 22 library adapters, -M 35 / 25 / 20 / 1: the 45-bp
# adapters at k = 11, the 35/36-bp ones at k = 1, 2, 11, 12 go through the filter) -- default stride, stride 1, one-chunk stretches
# with a recheck list of 7 entries, long reads, pools of 3 slots --, and the command line against the reference binary.
export TGSF_DEBUG_KNOBS=1   # the test settings below are read only under this switch
cd $GRAFT_REPO_ROOT
TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.3 timeout 900 python tests/manual/fuzz_campaign.py 440000 441200 150 > gpurun_out/r4_fuzz_filter_default.txt 2>&1; tail -2 gpurun_out/r4_fuzz_filter_default.txt
TGSF_MID_FILTER=1 TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.2 timeout 700 python tests/manual/fuzz_campaign.py 450000 450800 150 > gpurun_out/r4_fuzz_filter_stride1.txt 2>&1; tail -2 gpurun_out/r4_fuzz_filter_stride1.txt
TGSF_RECHECK_CAP=7 TGSF_FLAT_PMIN=1 TGSF_FLAT_PMAX=4 TGSF_FLAT_F0=100 TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.2 timeout 700 python tests/manual/fuzz_campaign.py 460000 460800 150 > gpurun_out/r4_fuzz_filter_tiny_stretches.txt 2>&1; tail -2 gpurun_out/r4_fuzz_filter_tiny_stretches.txt
TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.3 TGSF_FUZZ_MEAN_LEN=70000 timeout 700 python tests/manual/fuzz_campaign.py 470000 470300 30 > gpurun_out/r4_fuzz_filter_long.txt 2>&1; tail -2 gpurun_out/r4_fuzz_filter_long.txt
TGSF_POOL_CAP=3 TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.2 timeout 700 python tests/manual/fuzz_campaign.py 480000 480500 150 > gpurun_out/r4_fuzz_filter_replay.txt 2>&1; tail -2 gpurun_out/r4_fuzz_filter_replay.txt
timeout 900 python tests/manual/live_campaign.py 122000 122300 60 > gpurun_out/r4_campaign_filter.txt 2>&1; tail -3 gpurun_out/r4_campaign_filter.txt
