#!/bin/bash
# round 3, GPU call Y: evidence batch 1 - every BASELINE config through bench.py, rocprofv3 kernel stats of the headline
# and of C2, MFMA-utilisation counters of the Ritz kernels (n = 100: S-in-LDS; n = 200: S-stationary)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r3y; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
for w in dense_M512_k20 lap2d_5pt_M1e6_k100 graph_M1e7_k200 lap2d_5pt_M1e7_k500 lap3d_7pt_M1e8_k200; do
  timeout -k 10 400 python bench.py --workload $w --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_head -o head -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_headline_under_rocprof.json 2> $O/bench_headline_under_rocprof.err); echo "prof headline rc=$?"
find $O/prof_head -name "*kernel_stats.csv" -exec cp {} $O/headline_kernel_stats.csv \;
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python3 $ROOT/bench.py --workload lap2d_5pt_M1e6_k100 --steps 3 --warmup 1 --no-partial --no-cpu-baseline > $O/bench_c2_under_rocprof.json 2> $O/bench_c2_under_rocprof.err); echo "prof c2 rc=$?"
find $O/prof_c2 -name "*kernel_stats.csv" -exec cp {} $O/c2_kernel_stats.csv \;
for w in lap2d_5pt_M1e6_k100 lap2d_5pt_M1e7_k200; do
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/pmc_mfma_$w -o p -- python3 $ROOT/bench.py --workload $w --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-profile > $O/pmc_mfma_$w.out 2> $O/pmc_mfma_$w.err); echo "pmc mfma $w rc=$?"
done
python3 tools/pmc_mfma.py $O/pmc_mfma_lap2d_5pt_M1e6_k100 > $O/pmc_mfma_util_c2.json; python3 tools/pmc_mfma.py $O/pmc_mfma_lap2d_5pt_M1e7_k200 > $O/pmc_mfma_util_headline.json
rm -rf $O/prof_head $O/prof_c2 $O/pmc_mfma_lap2d_5pt_M1e6_k100 $O/pmc_mfma_lap2d_5pt_M1e7_k200
O=$O python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ["O"],"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    r=d.get("ritz_backtransform",{})
    print(os.path.basename(f), d["value"], d["config"].get("spmv_kernel"), {k:(v["avg_us"],v["frac"]) for k,v in d["roofline_all"].items()}, r.get("ms"), r.get("frac"), r.get("shader_clock_mhz"), r.get("mfma_issue_utilisation_in_cycles"), (d.get("cpu_baseline") or {}).get("value"))
PY
grep -E "gemm|k_update|k_qtw|k_spmv" $O/headline_kernel_stats.csv | cut -c1-200 | head -8
grep -E "gemm|k_update|k_qtw|k_spmv" $O/c2_kernel_stats.csv | cut -c1-200 | head -8
cat $O/pmc_mfma_util_c2.json | head -40
