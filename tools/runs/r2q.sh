#!/bin/bash
# round 2, GPU call Q: S-stationary Ritz GEMM (tune 9=5) vs the default at the headline size
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2q; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_lanczos.py -m gpu -q -x -k "ritz_backtransform" > $O/pytest_ritz.log 2>&1; rc=$?; echo "pytest ritz rc=$rc"; tail -3 $O/pytest_ritz.log
[ $rc -eq 0 ] || { grep -E "^E" $O/pytest_ritz.log | head -20; exit $rc; }
for v in 0 5; do timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-partial --no-cpu-baseline --tune 9=$v > $O/bench_ritz$v.json 2> $O/bench_ritz$v.err; echo "bench ritz variant $v rc=$?"; done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2q"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d.get("ritz_backtransform"))
PY
