#!/bin/bash
# round 3, GPU call Q: what does the 3-D 7-point SpMV wait for?  PMC passes of tools/spmv_pmc_probe.py (2-D vs 3-D stencils)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
python3 tools/spmv_pmc_probe.py > $O/plain.jsonl 2> $O/plain.err; echo "plain rc=$?"; cat $O/plain.jsonl
(cd /tmp && rocprofv3 -L > $O/counters_avail.txt 2>&1); echo "list rc=$?"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TD_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -o p -- python3 $ROOT/tools/spmv_pmc_probe.py > $O/pass_$i.out 2> $O/pass_$i.err); echo "pass $i [$set] rc=$?"
done
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.environ.get("O") or os.path.join(os.getcwd(), "gpurun_out", "r3q")
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
print(json.dumps(out, indent=1)[:6000])
PY
rm -rf $O/pmc_*
