#!/bin/bash
# round 2, GPU call V: MFMA utilisation counters of the Ritz GEMM kernels (new S-stationary default and the old kernel)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2v; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1; do
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_mfma_$v -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-profile --tune 9=$v > $O/pmc_mfma_$v.out 2> $O/pmc_mfma_$v.err); echo "pmc mfma variant $v rc=$?"
done
python3 tools/pmc_mfma.py $O/pmc_mfma_0 $O/pmc_mfma_1 > $O/pmc_mfma_util.json; cat $O/pmc_mfma_util.json
rm -rf $O/pmc_mfma_0 $O/pmc_mfma_1
