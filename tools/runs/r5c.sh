#!/bin/bash
# round 5, evidence call C: BASELINE C4 on one GPU (160 GB basis): bench line incl. the class-surface record
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python bench.py --workload lap3d_7pt_M1e8_k200 --steps 2 --warmup 1 --arm-timeout 900 > $O/bench_lap3d_7pt_M1e8_k200.json 2> $O/bench_lap3d_7pt_M1e8_k200.err; echo "bench C4 rc=$?"
tail -5 $O/bench_lap3d_7pt_M1e8_k200.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r5c/bench_lap3d_7pt_M1e8_k200.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k:(v["avg_us"],v["frac"]) for k,v in d["roofline_all"].items()}, d["ritz_backtransform"], (d.get("partial_reorth") or {}).get("ms_per_solve"), d.get("class_surface"), d.get("arm_error"))
PY
