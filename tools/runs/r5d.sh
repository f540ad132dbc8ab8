#!/bin/bash
# round 5, evidence call D: PMC HBM-traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the workloads given as arguments
# -> gpurun_out/r5d/hbm_traffic.json (a refreshed copy of profiles/hbm_traffic.json, every entry stamped)   
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
export LZ_TRAFFIC_STAMP="round 5, final tree"
[ -f $O/hbm_traffic.json ] || cp $ROOT/profiles/hbm_traffic.json $O/hbm_traffic.json
for w in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${w}_$c -o p -- python3 $ROOT/bench.py --workload $w --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-profile --no-class-surface > $O/pmc_${w}_$c.out 2> $O/pmc_${w}_$c.err); echo "pmc $w $c rc=$?"
  done
  python3 tools/make_traffic.py $w $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE $O/hbm_traffic.json > $O/traffic_$w.txt 2>&1; tail -30 $O/traffic_$w.txt
  rm -rf $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE
done
