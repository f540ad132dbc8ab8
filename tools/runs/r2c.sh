#!/bin/bash
# round 2, GPU call C: kernel trace of the C3 solve (two-phase SpMV kernels), Ritz GEMM A/B, one-reduce + distributed tests
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python3 $ROOT/bench.py --workload graph_M1e7_k200 --steps 1 --warmup 0 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err); echo "prof c3 rc=$?"
python3 - <<'PY' > $O/c3_kernel_stats.txt
import csv, glob, os
for f in glob.glob(os.environ.get("O", "gpurun_out/r2c") + "/prof_c3/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f)))[:14]:
        print(row["Name"][:70], row["Calls"], row["AverageNs"], row["Percentage"])
PY
cat $O/c3_kernel_stats.txt
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-partial > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout -k 10 600 python -m pytest tests/test_gpu_lanczos.py tests/test_gpu_kernels.py tests/test_gpu_distributed.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
