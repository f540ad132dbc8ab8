#!/bin/bash
# round 2, GPU call J: whole -m gpu suite, headline kernel trace (CSV stats), bench lines of every BASELINE config
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -22 $O/pytest.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o h -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-partial --no-prewarm > $O/bench_headline_under_rocprof.json 2> $O/bench_headline_under_rocprof.err); echo "prof headline rc=$?"
find $O/prof_headline -name "*kernel_stats.csv" -exec cp {} $O/headline_kernel_stats.csv \; ; head -8 $O/headline_kernel_stats.csv
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
for w in dense_M512_k20 lap2d_5pt_M1e6_k100 graph_M1e7_k200 lap2d_5pt_M1e7_k500; do
  timeout -k 10 400 python bench.py --workload $w --steps 3 --warmup 1 --no-partial > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
timeout -k 10 500 python bench.py --workload lap3d_7pt_M1e8_k200 --steps 2 --warmup 1 --no-partial --no-cpu-baseline > $O/bench_lap3d_7pt_M1e8_k200.json 2> $O/bench_lap3d_7pt_M1e8_k200.err; echo "bench c4 rc=$?"
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2j"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("spmv_kernel"), {k:(v["avg_us"],v["frac"]) for k,v in d["roofline_all"].items()}, d["ritz_backtransform"]["ms"], d["ritz_backtransform"]["frac"], (d.get("cpu_baseline") or {}).get("value"))
PY
