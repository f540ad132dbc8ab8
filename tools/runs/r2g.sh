#!/bin/bash
# round 2, GPU call G: two-phase SpMV v4 (perm through LDS), Ritz default = variant 1, steady-state Ritz timing
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lanczos.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/pytest.log; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -E "^E|FAILED" $O/pytest.log | head -20; exit $rc; }
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python3 $ROOT/bench.py --workload graph_M1e7_k200 --steps 1 --warmup 0 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err); echo "prof c3 rc=$?"
python3 tools/rocpd_stats.py $O/prof_c3 > $O/c3_kernel_stats.txt; head -8 $O/c3_kernel_stats.txt
for cap in 4096 10240; do timeout -k 10 300 python bench.py --workload graph_M1e7_k200 --steps 2 --warmup 1 --no-partial --no-cpu-baseline --tune 10=$cap > $O/bench_c3_cap$cap.json 2> $O/bench_c3_cap$cap.err; echo "bench c3 cap $cap rc=$?"; done
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-partial > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2g"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["config"].get("spmv_kernel"), {k:v["avg_us"] for k,v in d["roofline_all"].items()}, d["ritz_backtransform"])
PY
