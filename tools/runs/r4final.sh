#!/bin/bash
# round 4, evidence call: full GPU suite, smoke, driver-style + default bench lines, warmed rocprofv3 kernel stats of the headline
# and of the partial loop, PMC passes (HBM traffic of the headline and of C3, MFMA utilisation of the Ritz / Gram kernels), the
# other BASELINE configs.  Everything lands under gpurun_out/r4final/ (copied into profiles/r04/ afterwards).
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r4final; mkdir -p $O
export TMPDIR=/tmp
STAMP="round 4 (PMC passes of $(date +%Y-%m-%d), final tree of the round)"
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu_full_suite.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu_full_suite.log
[ $rc -eq 0 ] || { grep -E "^E" $O/pytest_gpu_full_suite.log | head -20; exit $rc; }
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench driver-style rc=$?"
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
# warmed kernel stats (the tracked csv must reproduce the line's frac): one warm-up solve + the default pre-warm
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline_stats -o p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-class-surface > $O/bench_headline_under_rocprof.json 2> $O/bench_headline_under_rocprof.err); echo "headline stats rc=$?"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/partial_stats -o p -- python3 $ROOT/tools/partial_probe.py --reps 3 --no-profile > $O/partial_probe_under_rocprof.jsonl 2> $O/partial_probe_under_rocprof.err); echo "partial stats rc=$?"
timeout -k 10 200 python tools/partial_probe.py --arms ";18=3;18=2;18=1" --reps 4 > $O/partial_probe.jsonl 2> $O/partial_probe.err; echo "partial probe rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_headline_$c -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-class-surface > $O/pmc_headline_$c.out 2> $O/pmc_headline_$c.err); echo "pmc headline $c rc=$?"
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_c3_$c -o p -- python3 $ROOT/bench.py --workload graph_M1e7_k200 --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-class-surface > $O/pmc_c3_$c.out 2> $O/pmc_c3_$c.err); echo "pmc c3 $c rc=$?"
done
cp profiles/hbm_traffic.json $O/hbm_traffic.json
LZ_TRAFFIC_STAMP="$STAMP" python3 tools/make_traffic.py lap2d_5pt_M1e7_k200 $O/pmc_headline_FETCH_SIZE $O/pmc_headline_WRITE_SIZE $O/hbm_traffic.json > $O/traffic_headline.out
LZ_TRAFFIC_STAMP="$STAMP" python3 tools/make_traffic.py graph_M1e7_k200 $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE $O/hbm_traffic.json > $O/traffic_c3.out
python3 tools/pmc_traffic.py $O/pmc_headline_FETCH_SIZE $O/pmc_headline_WRITE_SIZE > $O/headline_kernel_traffic.json
rm -rf $O/pmc_headline_* $O/pmc_c3_*
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-class-surface --no-profile > $O/pmc_mfma.out 2> $O/pmc_mfma.err); echo "pmc mfma rc=$?"
python3 tools/pmc_mfma.py $O/pmc_mfma > $O/pmc_mfma_util_headline.json; rm -rf $O/pmc_mfma
for w in dense_M512_k20 lap2d_5pt_M1e6_k100 graph_M1e7_k200 lap2d_5pt_M1e7_k500; do
  timeout -k 10 400 python bench.py --workload $w --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
timeout -k 10 500 python bench.py --workload lap3d_7pt_M1e8_k200 --steps 1 --warmup 1 --no-prewarm > $O/bench_lap3d_7pt_M1e8_k200.json 2> $O/bench_lap3d_7pt_M1e8_k200.err; echo "bench c4 rc=$?"
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4final/bench_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    p=d.get("partial_reorth") or {}
    print(os.path.basename(f), d["value"], d["ms_per_step"], {k:(v["avg_us"],v["frac"]) for k,v in d["roofline_all"].items()}, d["roofline"]["kernel"], d["roofline"]["frac"], (d.get("ritz_backtransform") or {}).get("ms"), (d.get("ritz_gram") or {}).get("ms"), p.get("ms_per_solve"), p.get("whole_iteration_frac_hbm_peak"), (d.get("cpu_baseline") or {}).get("value"))
PY
