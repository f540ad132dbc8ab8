#!/bin/bash
# round 2, GPU call B: re-run of the fixed tests + two-phase SpMV parity, teardown probe (fixed order), 2-rank benches, C3 bench
set -o pipefail
O=gpurun_out/r2b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_lanczos.py tests/test_gpu_kernels.py tests/test_gpu_two_sided.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
P=29517
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $P tools/teardown_probe.py --order torch_first --rccl > $O/probe_torch_first.out 2> $O/probe_torch_first.err; echo "probe torch_first rc=$?" | tee $O/probe_torch_first.rc
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $((P+1)) bench.py --gpus 2 --bootstrap torch --backend host --device 0 --workload lap2d_5pt_M1e6_k100 --steps 2 --warmup 1 --no-partial > $O/bench_2rank_torch.json 2> $O/bench_2rank_torch.err; echo "bench 2-rank torch bootstrap rc=$?" | tee $O/bench_2rank_torch.rc
timeout -k 10 300 python bench.py --gpus 2 --backend host --device 0 --workload lap2d_5pt_M1e6_k100 --steps 2 --warmup 1 --no-partial > $O/bench_2rank_spawn.json 2> $O/bench_2rank_spawn.err; echo "bench --gpus 2 (self-spawned ranks) rc=$?" | tee $O/bench_2rank_spawn.rc
timeout -k 10 300 python bench.py --workload graph_M1e7_k200 --steps 3 --warmup 1 --no-partial --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?"
timeout -k 10 300 python bench.py --workload graph_M1e7_k200 --steps 3 --warmup 1 --no-partial --no-cpu-baseline --tune 14=1 > $O/bench_c3_stream.json 2> $O/bench_c3_stream.err; echo "bench c3 (CSR-stream arm) rc=$?"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-partial --tune 9=1 > $O/bench_ritz_old.json 2> $O/bench_ritz_old.err; echo "bench (old ritz gemm) rc=$?"
tail -3 $O/pytest.log
