#!/bin/bash
# round 3, GPU call Z: final tree - full GPU suite, headline + C2 bench lines, C2 kernel stats and MFMA counters of the final Ritz kernel
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r3z; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu_full_suite.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gpu_full_suite.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout -k 10 300 python bench.py --workload lap2d_5pt_M1e6_k100 --steps 5 --warmup 2 > $O/bench_lap2d_5pt_M1e6_k100.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python3 $ROOT/bench.py --workload lap2d_5pt_M1e6_k100 --steps 3 --warmup 1 --no-partial --no-cpu-baseline > $O/bench_c2_under_rocprof.json 2> $O/bench_c2_under_rocprof.err); echo "prof c2 rc=$?"
find $O/prof_c2 -name "*kernel_stats.csv" -exec cp {} $O/c2_kernel_stats.csv \;
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_mfma_c2 -o p -- python3 $ROOT/bench.py --workload lap2d_5pt_M1e6_k100 --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-profile > $O/pmc_mfma_c2.out 2> $O/pmc_mfma_c2.err); echo "pmc mfma c2 rc=$?"
python3 tools/pmc_mfma.py $O/pmc_mfma_c2 > $O/pmc_mfma_util_c2.json
rm -rf $O/prof_c2 $O/pmc_mfma_c2
O=$O python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ["O"],"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    r=d.get("ritz_backtransform",{})
    print(os.path.basename(f), d["value"], {k:(v["avg_us"],v["frac"]) for k,v in d["roofline_all"].items()}, r.get("ms"), r.get("frac"), r.get("shader_clock_mhz"), r.get("mfma_issue_utilisation_in_cycles"), (d.get("cpu_baseline") or {}).get("value"))
PY
grep -E "gemm" $O/c2_kernel_stats.csv | cut -c1-220 | head -4
cat $O/pmc_mfma_util_c2.json | grep -A5 sl2
