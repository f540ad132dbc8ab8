#!/bin/bash
# round 5: what binds the row-class coded SpMV?  PMC passes over tools/spmv_coding_probe.py (headline matrix and the 300^3 grid), per kernel
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/pmc_spmv; mkdir -p $O
export TMPDIR=/tmp
export LZ_CASES="lap2d_5pt_4000x2500,lap3d_7pt_300^3"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 $ROOT/tools/spmv_coding_probe.py > $O/p$i.out 2> $O/p$i.err); echo "pass $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_spmv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_spmv" not in n: continue
        key = n.split("(")[0].replace("void ", "").replace("lz::", "") + " grid=" + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open("gpurun_out/pmc_spmv/summary.json", "w"), indent=1)
for k, cs in sorted(out.items()):
    print(k); print("   ", cs)
PY
rm -rf $O/p1 $O/p2 $O/p3
