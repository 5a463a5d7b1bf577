#!/bin/bash
# The campaign sittings of round 4 (gpurun from the repo root; logs under gpurun_out/, the judged copies in profiles/): the library
# against the oracle (tests/fuzz.py: random parameter sets, every record and tally word) on the new middle scan -- the default
# schedule, tiny stretches (every read cut into many stretches, stretches across read boundaries), pools of 3 slots (every batch
# through the replay), long reads --, and the command line against the reference binary.
export TGSF_DEBUG_KNOBS=1   # the test settings below are read only under this switch
cd $GRAFT_REPO_ROOT
TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.4 timeout 900 python tests/manual/fuzz_campaign.py 400000 401200 150 > gpurun_out/r4_fuzz_default.txt 2>&1; tail -2 gpurun_out/r4_fuzz_default.txt
TGSF_FLAT_PMIN=1 TGSF_FLAT_PMAX=4 TGSF_FLAT_F0=100 TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.3 timeout 900 python tests/manual/fuzz_campaign.py 410000 411000 150 > gpurun_out/r4_fuzz_tiny_stretches.txt 2>&1; tail -2 gpurun_out/r4_fuzz_tiny_stretches.txt
TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.5 TGSF_FUZZ_MEAN_LEN=70000 timeout 700 python tests/manual/fuzz_campaign.py 420000 420300 30 > gpurun_out/r4_fuzz_long.txt 2>&1; tail -2 gpurun_out/r4_fuzz_long.txt
TGSF_POOL_CAP=3 TGSF_REP_MAX_PLOG=1 TGSF_FUZZ_WIDE=1 TGSF_FUZZ_GATE_P=0.6 timeout 700 python tests/manual/fuzz_campaign.py 430000 430500 150 > gpurun_out/r4_fuzz_replay_and_memory_gate.txt 2>&1; tail -2 gpurun_out/r4_fuzz_replay_and_memory_gate.txt
timeout 900 python tests/manual/live_campaign.py 120000 120400 60 > gpurun_out/r4_campaign_a.txt 2>&1; tail -3 gpurun_out/r4_campaign_a.txt
TGSF_FLAT_PMIN=1 TGSF_FLAT_PMAX=8 TGSF_BATCH_BYTES=40000 TGSF_FILL_MIN_BYTES=1 TGSF_POOL_CAP=3 timeout 700 python tests/manual/live_campaign.py 121000 121300 40 > gpurun_out/r4_campaign_b.txt 2>&1; tail -3 gpurun_out/r4_campaign_b.txt
