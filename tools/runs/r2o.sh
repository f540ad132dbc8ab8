#!/bin/bash
# round 2, GPU call O: two-phase SpMV v7 (pairs + per-group destinations in phase 1; persistent phase 2 with register prefetch)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2o; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $O/pytest_kernels.log 2>&1; rc=$?; echo "pytest kernels rc=$rc"; tail -3 $O/pytest_kernels.log
[ $rc -eq 0 ] || { grep -E "^E" $O/pytest_kernels.log | head; exit $rc; }
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python3 $ROOT/bench.py --workload graph_M1e7_k200 --steps 1 --warmup 0 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err); echo "prof c3 rc=$?"
find $O/prof_c3 -name "*kernel_stats.csv" -exec cp {} $O/c3_two_phase_kernel_stats.csv \; ; head -7 $O/c3_two_phase_kernel_stats.csv | cut -c1-150
for cap in 0 10240 8192; do timeout -k 10 300 python bench.py --workload graph_M1e7_k200 --steps 2 --warmup 1 --no-partial --no-cpu-baseline --tune 10=$cap > $O/bench_c3_cap$cap.json 2> $O/bench_c3_cap$cap.err; echo "bench c3 cap $cap rc=$?"; done
timeout -k 10 400 python tools/ablate_r2.py pb > $O/ablate.json 2> $O/ablate.err; cat $O/ablate.json
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2o"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["config"].get("spmv_kernel"), {k:v["avg_us"] for k,v in d["roofline_all"].items()})
PY
