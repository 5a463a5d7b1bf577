#!/bin/bash
# The GPU sittings behind a round's profiles/ (run through gpurun from the repo root; outputs under gpurun_out/):
#   tools/round_end_gpu.sh profile <tag>   tools/profile_round.sh + gated profiles + kernel-path variants
#   tools/round_end_gpu.sh bench           the full default bench line (as the driver runs it) + --config c5
#   tools/round_end_gpu.sh suite           the GPU test suite
export TGSF_DEBUG_KNOBS=1   # the test settings below are read only under this switch
cd $GRAFT_REPO_ROOT
what=${1:-suite}; tag=${2:-r04}
R=$GRAFT_REPO_ROOT
brief() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('value %.1f Gbases/s  ms/step %.3f  sum_kernel_ms %.3f  critical_path_ms %.3f  frac %.4f' % (j['value'], j['ms_per_step'], r['sum_kernel_ms'], r['critical_path_ms'], r['frac']))
print({k: round(v,3) for k,v in r['stage_ms_per_step'].items()})"; }
case $what in
suite)
  python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/gpu_suite.txt 2>&1; tail -4 gpurun_out/gpu_suite.txt ;;
profile)
  bash tools/profile_round.sh $tag > gpurun_out/profile_round.log 2>&1; tail -2 gpurun_out/profile_round.log
  cp gpurun_out/prof_$tag/traffic.json profiles/traffic.json      # (on the box: later bench lines quote the traffic of THIS build)
  cd /tmp && export TMPDIR=/tmp
  for k in 11 21; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gate$k -- python3 $R/bench.py --no-e2e --no-cpu-baseline --no-oracle-check --min-repeat 100 --kmer $k --streams 1 --kernel-steps 6 > $R/gpurun_out/gate${k}_single.json 2> $R/gpurun_out/gate$k.err
  done
  cd $R
  for k in 11 12 13 16 31; do python bench.py --no-e2e --no-cpu-baseline --min-repeat 100 --kmer $k --kernel-steps 6 > gpurun_out/b_repeat_k$k.json 2> gpurun_out/b_repeat.err; done
  python bench.py --no-e2e --no-cpu-baseline --kernel-steps 12 > gpurun_out/b_nogate.json 2>> gpurun_out/b_repeat.err
  for v in "" "TGSF_NO_HOT32=1"; do echo "== --short-adapters $v"; env $v python bench.py --no-e2e --no-cpu-baseline --short-adapters --kernel-steps 12 2>/dev/null | brief; done > gpurun_out/short_adapters.txt 2>&1
  for v in "" "TGSF_STATS_NT=0" "TGSF_SEG_COLS=2048" "TGSF_SEG_COLS=4096"; do echo "== $v"; env $v python bench.py --no-e2e --no-cpu-baseline --no-oracle-check --kernel-steps 12 --streams 1 2>/dev/null | brief; done > gpurun_out/kernel_knobs.txt 2>&1
  cat gpurun_out/short_adapters.txt gpurun_out/kernel_knobs.txt ;;
bench)
  python bench.py --config c5 --steps 2 --warmup 1 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; tail -c 600 gpurun_out/bench_c5.json
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json ;;
esac
