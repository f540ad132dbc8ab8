#!/bin/bash
# round 2, GPU call Y: C4 (M = 1e8, 160 GB basis): pass-1 slice length and update-kernel shape against the defaults
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2y; mkdir -p $O
export TMPDIR=/tmp
for t in "0=0" "0=1024" "0=2048" "8=3" "8=4"; do
  timeout -k 10 280 python bench.py --workload lap3d_7pt_M1e8_k200 --steps 1 --warmup 1 --no-partial --no-cpu-baseline --no-prewarm --tune $t > $O/bench_c4_$t.json 2> $O/bench_c4_$t.err; echo "bench c4 $t rc=$?"
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2y"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["ms_per_step"], {k:(v["avg_us"],v["frac"]) for k,v in d["roofline_all"].items()})
PY
