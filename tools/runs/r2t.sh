#!/bin/bash
# round 2, GPU call T: full GPU suite + refreshed evidence (headline bench + kernel stats, C3 bench + PMC traffic, ablations)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2t; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 850 python -m pytest tests -m gpu -q -x --durations=12 > $O/pytest_gpu_full_suite.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu_full_suite.log
[ $rc -eq 0 ] || { grep -E "^E" $O/pytest_gpu_full_suite.log | head -20; exit $rc; }
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_head -o head -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_headline_under_rocprof.json 2> $O/bench_headline_under_rocprof.err); echo "prof headline rc=$?"
find $O/prof_head -name "*kernel_stats.csv" -exec cp {} $O/headline_kernel_stats.csv \;
timeout -k 10 400 python bench.py --workload graph_M1e7_k200 --steps 3 --warmup 1 --no-partial > $O/bench_graph_M1e7_k200.json 2> $O/bench_graph_M1e7_k200.err; echo "bench c3 rc=$?"
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python3 $ROOT/bench.py --workload graph_M1e7_k200 --steps 1 --warmup 0 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err); echo "prof c3 rc=$?"
find $O/prof_c3 -name "*kernel_stats.csv" -exec cp {} $O/c3_two_phase_kernel_stats.csv \;
cp $ROOT/profiles/hbm_traffic.json $O/hbm_traffic.json
w=graph_M1e7_k200
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${w}_$c -o p -- python3 $ROOT/bench.py --workload $w --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-profile > $O/pmc_${w}_$c.out 2> $O/pmc_${w}_$c.err); echo "pmc $w $c rc=$?"
done
python3 tools/make_traffic.py $w $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE $O/hbm_traffic.json | head -8
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("O", "gpurun_out/r2t")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_graph*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_pb_" in n:
            acc["k_pb_products" if "products" in n else "k_pb_rows" if "rows" in n else "k_pb_setup"][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v) * 1024) for c, v in cs.items()}, {c: len(v) for c, v in cs.items()})
PY
rm -rf $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE $O/prof_head $O/prof_c3
timeout -k 10 400 python tools/ablate_r2.py pb > $O/ablate_pb.json 2> $O/ablate_pb.err; cat $O/ablate_pb.json
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2t"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["config"].get("spmv_kernel"), {k:(v["avg_us"],v["frac"]) for k,v in d["roofline_all"].items()}, d.get("ritz_backtransform",{}).get("ms"), d.get("ritz_backtransform",{}).get("frac"))
PY
