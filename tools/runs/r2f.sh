#!/bin/bash
# round 2, GPU call F: small-problem engine (first run: short timeout around it), kbench ablations of k_pb_rows / GEMM
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 240 python -m pytest tests/test_gpu_small.py -m gpu -q -x -s > $O/pytest_small.log 2>&1; rc=$?; echo "pytest small rc=$rc"; grep -E "small-engine|passed|failed|Error" $O/pytest_small.log | head -30
[ $rc -eq 0 ] || { tail -30 $O/pytest_small.log; exit $rc; }
timeout -k 10 600 python tools/ablate_r2.py > $O/ablate.json 2> $O/ablate.err; echo "ablate rc=$?"; cat $O/ablate.json
