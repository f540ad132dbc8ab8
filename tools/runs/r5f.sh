#!/bin/bash
# round 5, evidence call F (after the row-class coded SpMV became the stencil default): rocprofv3 --stats of the headline bench and of the
# selective loop, PMC HBM-traffic passes of every stencil workload (their spmv entries changed format), then the bench lines re-taken
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
export LZ_TRAFFIC_STAMP="round 5, final tree (row-class coded SpMV)"
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_headline -o p -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-prewarm --no-cpu-baseline --no-class-surface > $O/bench_headline_under_rocprof.json 2> $O/bench_headline_under_rocprof.err); echo "rocprof headline rc=$?"
python3 tools/rocpd_stats.py $O/prof_headline csv > $O/headline_kernel_stats.csv 2> $O/rocpd.err || cp $(find $O/prof_headline -name "*kernel_stats.csv" | head -1) $O/headline_kernel_stats.csv
rm -rf $O/prof_headline
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_partial -o p -- python3 $ROOT/tools/partial_probe.py --reps 3 > $O/partial_probe_under_rocprof.jsonl 2> $O/partial_probe.err); echo "rocprof partial rc=$?"
python3 tools/rocpd_stats.py $O/prof_partial csv > $O/partial_loop_kernel_stats.csv 2>> $O/rocpd.err || cp $(find $O/prof_partial -name "*kernel_stats.csv" | head -1) $O/partial_loop_kernel_stats.csv
rm -rf $O/prof_partial
head -8 $O/headline_kernel_stats.csv | cut -c1-200; head -8 $O/partial_loop_kernel_stats.csv | cut -c1-200
cp $ROOT/profiles/hbm_traffic.json $O/hbm_traffic.json
for w in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${w}_$c -o p -- python3 $ROOT/bench.py --workload $w --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-profile --no-class-surface > $O/pmc_${w}_$c.out 2> $O/pmc_${w}_$c.err); echo "pmc $w $c rc=$?"
  done
  python3 tools/make_traffic.py $w $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE $O/hbm_traffic.json > $O/traffic_$w.txt 2>&1; grep -A4 '"spmv"' $O/traffic_$w.txt | head -6
  rm -rf $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE
done
cp $O/hbm_traffic.json $ROOT/profiles/hbm_traffic.json   # (on the box: the bench lines below read it; merged back by hand from gpurun_out/r5f)
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench driver-style rc=$?"
timeout -k 10 600 python bench.py --workload deuteron3d_N160_27pt_k400 --steps 3 --warmup 1 > $O/bench_deuteron3d_N160_27pt_k400.json 2> $O/bench_deuteron.err; echo "bench deuteron rc=$?"
timeout -k 10 300 python tools/partial_step_probe.py > $O/partial_step_probe.jsonl 2> $O/partial_step_probe.err; echo "partial step probe rc=$?"
for w in "$@"; do
  [ "$w" = lap2d_5pt_M1e7_k200 ] && continue
  timeout -k 10 900 python bench.py --workload $w --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5f/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    p = d.get("partial_reorth") or {}
    s = d["roofline_all"].get("spmv", {})
    print(f.split("/")[-1], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], "spmv", s.get("avg_us"), s.get("frac"), s.get("traffic"), s.get("bytes_per_launch"), "partial", p.get("ms_per_solve"), d["config"].get("spmv_coding"))
PY
