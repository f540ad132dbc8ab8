#!/bin/bash
# round 2, GPU call D: two-phase SpMV v2 + LDS-staged Ritz GEMM: tests, kernel trace of the C3 solve, benches
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_lanczos.py tests/test_gpu_kernels.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/pytest.log; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python3 $ROOT/bench.py --workload graph_M1e7_k200 --steps 1 --warmup 0 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err); echo "prof c3 rc=$?"
python3 tools/rocpd_stats.py $O/prof_c3 > $O/c3_kernel_stats.txt; head -12 $O/c3_kernel_stats.txt
timeout -k 10 300 python bench.py --workload graph_M1e7_k200 --steps 3 --warmup 1 --no-partial --no-cpu-baseline --tune 10=16384 > $O/bench_c3_cap16k.json 2> $O/bench_c3_cap16k.err; echo "bench c3 cap16k rc=$?"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-partial > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-partial --tune 9=2 > $O/bench_ritz_persist1.json 2> $O/bench_ritz_persist1.err; echo "bench ritz variant 2 rc=$?"
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2d"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["config"].get("spmv_kernel"), {k:v["avg_us"] for k,v in d["roofline_all"].items()}, d["ritz_backtransform"]["ms"], d["ritz_backtransform"]["frac"])
PY
