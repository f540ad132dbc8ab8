#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 pass:
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs) per kernel (gfx94x formula, see
MI355X_MICROARCH.md 'rocprofv3 PMC slots')."""
import collections, csv, glob, json, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    o = {c: sum(v) / len(v) for c, v in cs.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and o.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs
        o["mfma_util_pct"] = 100.0 * o["SQ_VALU_MFMA_BUSY_CYCLES"] / (o["GRBM_GUI_ACTIVE"] / 8.0 * 1024)
    if "mfma" in k or "gemm" in k or "gram" in k:
        out[k] = o
print(json.dumps(out, indent=1))
