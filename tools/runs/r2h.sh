#!/bin/bash
# round 2, GPU call H: engine with the write-through protocol, two-phase SpMV v5 + ablation arms
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2h; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 240 python -m pytest tests/test_gpu_small.py -m gpu -q -x -s > $O/pytest_small.log 2>&1; rc=$?; echo "pytest small rc=$rc"; grep -E "small-engine|passed|failed|Error" $O/pytest_small.log | head -30
[ $rc -eq 0 ] || { tail -30 $O/pytest_small.log; exit $rc; }
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $O/pytest_kernels.log 2>&1; rc=$?; echo "pytest kernels rc=$rc"; tail -3 $O/pytest_kernels.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/ablate_r2.py pb > $O/ablate.json 2> $O/ablate.err; echo "ablate rc=$?"; cat $O/ablate.json
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python3 $ROOT/bench.py --workload graph_M1e7_k200 --steps 1 --warmup 0 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err); echo "prof c3 rc=$?"
python3 tools/rocpd_stats.py $O/prof_c3 > $O/c3_kernel_stats.txt; head -6 $O/c3_kernel_stats.txt
