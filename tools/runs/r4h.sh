#!/bin/bash
# round 4, GPU call H: the GPU suite files touched since call F, C3 two-phase kernels' HBM traffic per kernel (PMC), bench lines
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_two_sided.py tests/test_gpu_lanczos.py tests/test_gpu_small.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -o p -- python3 $ROOT/tools/pb_once.py > $O/pb_once.out 2> $O/pb_once.err); echo "c3 stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/c3_pmc_$c -o p -- python3 $ROOT/tools/pb_once.py > $O/pb_pmc_$c.out 2> $O/pb_pmc_$c.err); echo "c3 pmc $c rc=$?"
done
python3 tools/pmc_traffic.py $O/c3_pmc_FETCH_SIZE $O/c3_pmc_WRITE_SIZE > $O/c3_pb_kernel_traffic.json
rm -rf $O/c3_pmc_*
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 400 python bench.py --workload graph_M1e7_k200 --steps 3 --warmup 1 --no-cpu-baseline --no-class-surface > $O/bench_graph_M1e7_k200.json 2> $O/bench_graph.err; echo "bench c3 rc=$?"
