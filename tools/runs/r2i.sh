#!/bin/bash
# round 2, GPU call I: PMC counters of the two-phase SpMV kernels (what is phase 2 waiting for?)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -o p -- python3 $ROOT/tools/pb_once.py > $O/pmc_$tag.out 2> $O/pmc_$tag.err); echo "pmc $tag rc=$?"
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("O", "gpurun_out/r2i")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_pb_" not in k: continue
        acc["rows" if "k_pb_rows" in k else ("products" if "k_pb_products" in k else "other")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
