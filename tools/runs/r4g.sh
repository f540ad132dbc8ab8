#!/bin/bash
# round 4, GPU call G: (1) tests of the new pieces, (2) C3 with the constant-value two-phase stream: bench line, kernel stats,
# HBM traffic (PMC), (3) the fixed-K SpMV in its three layouts (CSR order / ELL / ELL two rows per lane): time + counters
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_two_sided.py tests/test_gpu_lanczos.py -x -q -m gpu -k "two_phase or spmv or gram or bireorth or dtype or float32 or checkpoint or error" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 400 python bench.py --workload graph_M1e7_k200 --steps 3 --warmup 1 --no-class-surface > $O/bench_graph_M1e7_k200.json 2> $O/bench_graph.err; echo "bench c3 rc=$?"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -o p -- python3 $ROOT/tools/pb_once.py > $O/pb_once.out 2> $O/pb_once.err); echo "c3 stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/c3_pmc_$c -o p -- python3 $ROOT/tools/pb_once.py > $O/pb_pmc_$c.out 2> $O/pb_pmc_$c.err); echo "c3 pmc $c rc=$?"
done
python3 tools/pmc_traffic.py $O/c3_pmc_FETCH_SIZE $O/c3_pmc_WRITE_SIZE > $O/c3_pb_kernel_traffic.json; cat $O/c3_pb_kernel_traffic.json | head -60
for lay in 1 2 3; do
  LZ_LAYOUT=$lay timeout -k 10 200 python3 tools/spmv_pmc_probe.py > $O/spmv_plain_layout$lay.jsonl 2> $O/spmv_plain_layout$lay.err; echo "plain layout $lay rc=$?"; cat $O/spmv_plain_layout$lay.jsonl
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TA_BUSY_avr TD_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  for lay in 1 2; do
    (cd /tmp && LZ_LAYOUT=$lay timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${lay}_$i -o p -- python3 $ROOT/tools/spmv_pmc_probe.py > $O/pass_${lay}_$i.out 2> $O/pass_${lay}_$i.err); echo "pass $i layout $lay [$set] rc=$?"
  done
done
O=$O python3 - <<'PY'
import csv, glob, json, os, collections
O = os.environ["O"]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "spmv" not in k:
            continue
        res[(k.split("(")[0], row.get("Grid_Size"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for (k, g), d in sorted(res.items()):
    out[f"{k} grid={g}"] = {c: (sum(v) / len(v)) for c, v in d.items()}
json.dump(out, open(O + "/spmv_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -rf $O/pmc_* $O/c3_pmc_*
