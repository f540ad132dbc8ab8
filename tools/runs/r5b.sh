#!/bin/bash
# round 5, evidence call B: the other BASELINE configs + the reference's largest run + the dense workloads, one bench line each;
# rocprofv3 kernel stats of the GEMV-bound dense workload
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
for w in lap2d_5pt_M1e6_k100 graph_M1e7_k200 lap2d_5pt_M1e7_k500 deuteron3d_N160_27pt_k400 dense_M32768_k100 dense_M512_k20; do
  timeout -k 10 600 python bench.py --workload $w --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_dense -o p -- python3 $ROOT/bench.py --workload dense_M32768_k100 --steps 3 --warmup 1 --no-prewarm --no-cpu-baseline --no-class-surface --no-partial > $O/bench_dense_under_rocprof.json 2> $O/bench_dense_under_rocprof.err); echo "rocprof dense rc=$?"
python3 tools/rocpd_stats.py $O/prof_dense csv > $O/dense_M32768_kernel_stats.csv 2> $O/rocpd.err || cp $(find $O/prof_dense -name "*kernel_stats.csv" | head -1) $O/dense_M32768_kernel_stats.csv
head -8 $O/dense_M32768_kernel_stats.csv
rm -rf $O/prof_dense
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r5b/bench_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    p=d.get("partial_reorth") or {}
    print(os.path.basename(f), d["value"], d["ms_per_step"], {k:(v["avg_us"],v["frac"]) for k,v in d["roofline_all"].items()}, "ritz", d["ritz_backtransform"]["ms"], d["ritz_backtransform"]["frac"],
          "gram", (d.get("ritz_gram") or {}).get("ms"), (d.get("ritz_gram") or {}).get("frac"), "partial", p.get("ms_per_solve"), p.get("sweeps"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
