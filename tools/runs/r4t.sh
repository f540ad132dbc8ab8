#!/bin/bash
# round 4, GPU call T: per-kernel durations of the grouped symmetric Gram kernels at C5 (n = 500)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r4t; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $ROOT/bench.py --workload lap2d_5pt_M1e7_k500 --steps 1 --warmup 0 --no-prewarm --no-cpu-baseline --no-class-surface --no-partial > $O/bench.json 2> $O/bench.err); echo "rc=$?"
grep -E "gram|gemm|sum_slices" $O/stats/p_kernel_stats.csv | cut -c1-60,200-400 | head
grep -E "gram_unit" $O/stats/p_kernel_stats.csv | awk -F'",' '{print $2}' | head
