#!/bin/bash
# is the C2 partial arm's 10.9 ms (r5b) reproducible, and does it depend on the VMM-backed basis?
ROOT=$(pwd); O=$ROOT/gpurun_out/c2q; mkdir -p $O
for i in 1 2 3; do python bench.py --workload lap2d_5pt_M1e6_k100 --steps 3 --warmup 1 --no-cpu-baseline --no-class-surface > $O/b$i.json 2> $O/b$i.err; done
LZ_NO_VMM=1 python bench.py --workload lap2d_5pt_M1e6_k100 --steps 3 --warmup 1 --no-cpu-baseline --no-class-surface > $O/b_novmm.json 2> $O/b_novmm.err
LZ_DEBUG_TIMING=1 python bench.py --workload lap2d_5pt_M1e6_k100 --steps 3 --warmup 1 --no-cpu-baseline --no-class-surface > $O/b_dbg.json 2> $O/b_dbg.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c2q/b*.json")):
    d=json.load(open(f)); p=d["partial_reorth"]
    print(f, d["value"], p["ms_per_solve"], p["device_ms_per_solve"], p["spmv_share_of_device_time"], d["ritz_gram"]["ms"])
PY
tail -12 $O/b_dbg.err
