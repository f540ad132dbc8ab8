#!/bin/bash
# C3 two-phase SpMV vs Infinity-Cache residency of its product stream: timings + PMC traffic per size
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5mall
mkdir -p $O
cd $R
python3 tools/pb_mall_probe.py > $O/timings.jsonl 2> $O/timings.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c --output-format csv -- python3 tools/pb_mall_probe.py 1e6 2e6 4e6 > $O/pmc_$c.log 2>&1
done
python3 tools/pmc_by_size.py $O > $O/traffic_by_size.json
cat $O/timings.jsonl; cat $O/traffic_by_size.json
