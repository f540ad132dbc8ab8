#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files into HBM bytes per launch per kernel.
usage: pmc_traffic.py <dir with *_counter_collection.csv> ...   (FETCH_SIZE doubled on gfx950, see MI355X_MICROARCH.md)"""
import csv, glob, json, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    o = {c: sum(v) / len(v) for c, v in cs.items()}
    o["launches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in o:
        o["read_bytes"] = 2 * o["FETCH_SIZE"] * 1024  # FETCH_SIZE is in KiB and counts 128-B requests as 64 B on gfx950
    if "WRITE_SIZE" in o:
        o["write_bytes"] = o["WRITE_SIZE"] * 1024
    out[k] = o
print(json.dumps(out, indent=1))
