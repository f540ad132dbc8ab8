import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(f"{k:14s}", {c: (round(x["us_med"], 1), round(x["GBps_med"] or 0)) for c, x in v.items() if isinstance(x, dict)}, v.get("c_maxdiff_vs_first"))
