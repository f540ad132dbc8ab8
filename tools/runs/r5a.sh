#!/bin/bash
# round 5, evidence call A: headline - driver-style + default bench lines, rocprofv3 kernel stats of whole solves, the partial loops'
# kernel stats (three-collective device loop and the one-reduce loop at the N = 8 rank share)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench driver-style rc=$?"
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_headline -o p -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-prewarm --no-cpu-baseline --no-class-surface > $O/bench_headline_under_rocprof.json 2> $O/bench_headline_under_rocprof.err); echo "rocprof headline rc=$?"
python3 tools/rocpd_stats.py $O/prof_headline csv > $O/headline_kernel_stats.csv 2> $O/rocpd.err || cp $(find $O/prof_headline -name "*kernel_stats.csv" | head -1) $O/headline_kernel_stats.csv
head -12 $O/headline_kernel_stats.csv
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_partial -o p -- python3 $ROOT/tools/partial_probe.py --reps 3 > $O/partial_probe_under_rocprof.jsonl 2> $O/partial_probe.err); echo "rocprof partial rc=$?"
python3 tools/rocpd_stats.py $O/prof_partial csv > $O/partial_loop_kernel_stats.csv 2>> $O/rocpd.err || cp $(find $O/prof_partial -name "*kernel_stats.csv" | head -1) $O/partial_loop_kernel_stats.csv
head -10 $O/partial_loop_kernel_stats.csv
rm -rf $O/prof_headline $O/prof_partial
timeout -k 10 300 python tools/collective_budget.py $O/collective_budget.json > $O/collective_budget.txt 2>&1; echo "collective budget rc=$?"
python3 - <<'PY'
import json
for f in ("bench_driver_style","bench_default"):
    d=json.loads(open("gpurun_out/r5a/%s.json"%f).read().strip().splitlines()[-1])
    p=d["partial_reorth"]; c=d["class_surface"]
    print(f, d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], "spmv", d["spmv_frac_hbm_peak"])
    print("  partial", p["ms_per_solve"], p["iterations_per_s"], p["host_syncs_inside_lz_run"], p["whole_iteration_frac_hbm_peak"])
    print("  ritz", d["ritz_backtransform"]["ms"], d["ritz_backtransform"]["frac"], "gram", d["ritz_gram"]["ms"], d["ritz_gram"]["frac"])
    print("  class", c["first_call"], c["second_call"]["overhead_s"], c["H_eigvals_s"], c["V_fetch_s"], c["H_eigvecs_fetch_s"])
    print("  cpu", d.get("cpu_baseline",{}).get("value"))
PY
