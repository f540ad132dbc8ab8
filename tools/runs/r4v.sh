#!/bin/bash
# round 4, GPU call V: HBM traffic (PMC) of the device-decided partial loop's kernels at the headline size
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r4v; mkdir -p $O
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 $ROOT/tools/partial_probe.py --reps 1 --no-profile > $O/probe_$c.out 2> $O/probe_$c.err); echo "pmc $c rc=$?"
done
python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/partial_loop_kernel_traffic.json
rm -rf $O/pmc_*
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4v/partial_loop_kernel_traffic.json"))
for k,v in d.items():
    if "lz::" in k: print(k[:60], v.get("launches"), round(v.get("read_bytes",0)/1e9,4), round(v.get("write_bytes",0)/1e9,4))
PY
