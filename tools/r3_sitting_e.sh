cd $GRAFT_REPO_ROOT
timeout 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k 'pool or overflow' --durations=8 > gpurun_out/r3_pool_tests2.txt 2>&1; tail -12 gpurun_out/r3_pool_tests2.txt
P=tgsfilter_amd/bin/tgsfilter_prev
python tests/manual/e2e_threads.py 400000 3 '' "CLI=$P" '' "CLI=$P" 'TGSF_DETACH=1' "CLI=$P" 'TGSF_NO_EARLY_RESERVE=1' 'TGSF_SCAN_THREADS=8' "CLI=$P TGSF_SCAN_THREADS=8" > gpurun_out/r3_threads_c.txt 2>&1; cat gpurun_out/r3_threads_c.txt
for v in "" "TGSF_SEG_COLS=2048" "TGSF_SEG_COLS=4096" "TGSF_STATS_NT=1" "TGSF_STATS_NT=1 TGSF_SEG_COLS=2048"; do
  echo "== $v"; env $v python bench.py --no-e2e --no-cpu-baseline --no-oracle-check --kernel-steps 12 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('value %.1f Gbases/s  ms/step %.3f  sum_kernel_ms %.3f crit %.3f' % (j['value'], j['ms_per_step'], r['sum_kernel_ms'], r['critical_path_ms']))
print({k: round(v,3) for k,v in r['stage_ms_per_step'].items()})"
done > gpurun_out/r3_kernel_knobs.txt 2>&1; cat gpurun_out/r3_kernel_knobs.txt
