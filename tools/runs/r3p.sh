#!/bin/bash
# round 3, GPU call P: C3 (two-phase SpMV with the diagonal split + fp32 value stream): kernel stats + PMC HBM traffic
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
w=graph_M1e7_k200
timeout -k 10 400 python bench.py --workload $w --steps 3 --warmup 1 --no-partial > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench c3 rc=$?"
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python3 $ROOT/bench.py --workload $w --steps 1 --warmup 0 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err); echo "prof c3 rc=$?"
find $O/prof_c3 -name "*kernel_stats.csv" -exec cp {} $O/c3_two_phase_kernel_stats.csv \;
cp $ROOT/profiles/hbm_traffic.json $O/hbm_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${w}_$c -o p -- python3 $ROOT/bench.py --workload $w --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-profile > $O/pmc_${w}_$c.out 2> $O/pmc_${w}_$c.err); echo "pmc $w $c rc=$?"
done
python3 tools/make_traffic.py $w $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE $O/hbm_traffic.json | head -12
O=$O python3 - <<'PY'
import csv, glob, os, collections, json
O = os.environ["O"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_graph*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_pb_" in n:
            acc["k_pb_products" if "products" in n else "k_pb_rows" if "rows" in n else "k_pb_setup"][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v) * 1024) for c, v in cs.items()} for k, cs in acc.items()}
print(json.dumps(out))
json.dump(out, open(O + "/c3_pb_kernel_counters.json", "w"), indent=1)
PY
rm -rf $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE $O/prof_c3
grep -E "k_pb_|k_qtw|k_update" $O/c3_two_phase_kernel_stats.csv | cut -c1-220 | head
