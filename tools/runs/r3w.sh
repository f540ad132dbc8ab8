#!/bin/bash
# round 3, GPU call W: HBM traffic (PMC) of the Ritz back-transform kernels at C2 size (n = 100: S-in-LDS) and n = 200 (S-stationary)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r3w; mkdir -p $O
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 $ROOT/tools/ritz_probe.py 1000 1000 -- 100:0 200:0 50:0 > $O/probe_$c.out 2> $O/probe_$c.err); echo "pmc $c rc=$?"
done
python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/ritz_kernel_traffic_all.json
O=$O python3 - <<'PY'
import json, os
O = os.environ["O"]
d = json.load(open(O + "/ritz_kernel_traffic_all.json"))
out = {}
for k, v in d.items():
    if "gemm" in k:
        out[k] = {"read_bytes(2xFETCH_SIZE)": v.get("read_bytes"), "write_bytes": v.get("write_bytes"), "launches": v.get("launches")}
M = 1e6
for k, v in out.items():
    import re
    m = re.search(r"<(\d+), (\d+)", k)
    print(k, v)
json.dump(out, open(O + "/ritz_kernel_traffic.json", "w"), indent=1)
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
