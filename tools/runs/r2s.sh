#!/bin/bash
# round 2, GPU call S: S-stationary Ritz GEMM with LDS-DMA loads and 16-byte result stores: tests, timing arms, bench
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2s; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_lanczos.py -m gpu -q -x -k "ritz_backtransform" > $O/pytest_ritz.log 2>&1; rc=$?; echo "pytest ritz rc=$rc"; tail -3 $O/pytest_ritz.log
[ $rc -eq 0 ] || { grep -E "^E" $O/pytest_ritz.log | head -20; exit $rc; }
timeout -k 10 600 python tools/ablate_r2.py sreg > $O/ablate_sreg.json 2> $O/ablate_sreg.err; grep -v "^k_gemm" $O/ablate_sreg.json; grep "^k_gemm" $O/ablate_sreg.json | awk "{print \$2, \$3, \$4, \$5, \$6, \$(NF-3), \$(NF-2), \$(NF-1), \$NF}" | sort | uniq -c | sort -k2,6 -k1nr | awk "{k=\$2\$3\$4\$5\$6; c[k]++; if (c[k]<=1) print}"
for v in 0 5; do timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-partial --no-cpu-baseline --tune 9=$v > $O/bench_ritz$v.json 2> $O/bench_ritz$v.err; echo "bench ritz variant $v rc=$?"; done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2s"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d.get("ritz_backtransform"))
PY
