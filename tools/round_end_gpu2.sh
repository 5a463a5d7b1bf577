cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 300 gpurun_out/bench_final.json
(timeout 600 tools/repeat_probe 2048; timeout 300 tools/repeat_probe 65536) > gpurun_out/repeat_probe_final.txt 2>&1
timeout 1200 python tests/manual/e2e_c5.py 100000 > gpurun_out/e2e_c5.txt 2>&1; tail -30 gpurun_out/e2e_c5.txt
TGSF_LIVE_GATE_P=0.7 timeout 1500 python tests/manual/live_campaign.py 50000 50150 150 > gpurun_out/campaign_r2g.txt 2>&1; tail -3 gpurun_out/campaign_r2g.txt
