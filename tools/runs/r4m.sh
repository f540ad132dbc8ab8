#!/bin/bash
# round 4, GPU call M: warmed rocprofv3 per-kernel stats of the headline bench WITHOUT the short pre-warm solves (their k = 12
# launches would drag the per-kernel averages): five whole solves, the first one a warm-up
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r4m; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline_stats -o p -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-prewarm --no-cpu-baseline --no-class-surface --no-partial > $O/bench_headline_under_rocprof.json 2> $O/bench_headline_under_rocprof.err); echo "headline stats rc=$?"
head -7 $O/headline_stats/p_kernel_stats.csv | cut -c1-220
python3 -c "
import json
d=json.loads(open('$O/bench_headline_under_rocprof.json').read().strip().splitlines()[-1]); print(d['value'], {k:(v['avg_us'],v['frac']) for k,v in d['roofline_all'].items()})"
