#!/usr/bin/env python3
"""rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0 --no-prewarm` -> profiles/hbm_traffic.json
(HBM bytes per launch per kernel class; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
A class may consist of several kernels per logical launch (the two-phase irregular SpMV = k_pb_products + k_pb_rows):
bytes per launch = sum over all kernels of the class / launches of its most frequent kernel.
usage: make_traffic.py <workload> <fetch_dir> <write_dir> [out.json]"""
import collections, csv, glob, json, os, sys

CLASS = {"k_spmv": "spmv", "k_gemv": "spmv", "k_pb_": "spmv", "k_qtw": "qtw", "k_update": "update", "k_three_term": "three_term"}


def per_class(d, counter):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].replace("void ", "").replace("lz::", "").replace("(anonymous namespace)::", "")
            for k, c in CLASS.items():
                if name.startswith(k):
                    acc[c][name.split("(")[0]].append(float(row["Counter_Value"]))
    out = {}
    for c, kernels in acc.items():
        launches = max(len(v) for v in kernels.values())
        out[c] = (sum(sum(v) for v in kernels.values()) / launches, launches)
    return out


workload, fdir, wdir = sys.argv[1:4]
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "hbm_traffic.json")
fetch, write = per_class(fdir, "FETCH_SIZE"), per_class(wdir, "WRITE_SIZE")
data = json.load(open(out)) if os.path.isfile(out) else {}
data[workload] = {c: round(2 * 1024 * fetch[c][0] + 1024 * write.get(c, (0, 0))[0]) for c in fetch}
data[workload + "_measured"] = os.environ.get("LZ_TRAFFIC_STAMP", "undated")  # which round / date / tree these passes were taken on
data[workload + "_detail"] = {c: {"read_bytes": round(2 * 1024 * fetch[c][0]), "write_bytes": round(1024 * write.get(c, (0, 0))[0]), "launches": fetch[c][1]} for c in fetch}
json.dump(data, open(out, "w"), indent=1)
print(json.dumps(data[workload + "_detail"], indent=1))
