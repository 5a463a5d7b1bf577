#!/bin/bash
# Where the keys kernel of the repeat gate spends its time: the kernel path with and without -p 100 on fragment-length
# distributions that stay on the single-pass path (every fragment <= 65 536 k-mers), on the passes by leading bases (most
# of the bases beyond), and on the bench's own shape.   tools/repeat_shapes.sh  (on the GPU box)
B="python bench.py --no-e2e --no-cpu-baseline --no-oracle-check --streams 1 --kernel-steps 4 --kernel-warmup 2"
run() {
  $B "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(d['kernel_path']['gbases_per_step'], sum(d['roofline']['stage_ms_per_step'].values()))"
}
SHAPES=("" "--max-len 60000" "--mean-len 30000 --max-len 45000" "--mean-len 100000 --max-len 120000 --reads 65536" "--mean-len 200000 --max-len 250000 --reads 32768")
[ -n "$BENCH_SHAPE_ONLY" ] && SHAPES=("")
for shape in "${SHAPES[@]}"; do
  read g base < <(run $shape)
  for k in ${KS:-11 13 16}; do
    read g2 gated < <(run $shape --min-repeat 100 --kmer $k)
    python -c "print('shape [%s] -k %d: %.2f Gbases a step, no gate %.2f ms, gate +%.2f ms = %.2f ms per Gbase (%.2fx)' % ('$shape', $k, $g, $base, $gated - $base, ($gated - $base) / $g, $gated / $base))"
  done
done
