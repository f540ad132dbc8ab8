#!/bin/bash
# round 5, last GPU call: the full GPU suite + smoke on the final tree
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r5z; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu_full_suite.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gpu_full_suite.log
[ $rc -eq 0 ] || { grep -E "^E" $O/pytest_gpu_full_suite.log | head -20; exit $rc; }
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
