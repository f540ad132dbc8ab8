#!/bin/bash
# round 2, GPU call E: two-phase SpMV v3, stencil block builder + device potential, one-reduce, FP64 MFMA ceiling probe
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 120 tools/probes/mfma_f64_peak 20000 > $O/mfma_f64_peak.jsonl 2> $O/mfma_f64_peak.err; echo "mfma probe rc=$?"; cat $O/mfma_f64_peak.jsonl
timeout -k 10 700 python -m pytest tests/test_gpu_kernels.py tests/test_hamiltonian.py tests/test_gpu_distributed.py -m gpu -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/pytest.log; tail -15 $O/pytest.log
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python3 $ROOT/bench.py --workload graph_M1e7_k200 --steps 1 --warmup 0 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err); echo "prof c3 rc=$?"
python3 tools/rocpd_stats.py $O/prof_c3 > $O/c3_kernel_stats.txt; head -8 $O/c3_kernel_stats.txt
timeout -k 10 300 python bench.py --workload graph_M1e7_k200 --steps 3 --warmup 1 --no-partial --no-cpu-baseline --tune 10=16384 > $O/bench_c3_cap16k.json 2> $O/bench_c3_cap16k.err; echo "bench c3 cap16k rc=$?"
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2e"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["config"].get("spmv_kernel"), {k:v["avg_us"] for k,v in d["roofline_all"].items()}, d["ritz_backtransform"]["ms"], d["ritz_backtransform"]["frac"])
PY
