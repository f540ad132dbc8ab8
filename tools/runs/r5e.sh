#!/bin/bash
# round 5, evidence call E: C2 and C3 bench lines re-taken AFTER the PMC traffic refresh (their `roofline.traffic_source` stamps), and the
# Gram probe on the final tree
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
for w in lap2d_5pt_M1e6_k100 graph_M1e7_k200; do
  timeout -k 10 600 python bench.py --workload $w --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
timeout -k 10 300 python tools/gram_sizes_probe.py c2,headline,n400,c5 > $O/gram_sizes_final.jsonl 2>&1; cat $O/gram_sizes_final.jsonl | cut -c1-230
python3 - <<'PY'
import json
for w in ("lap2d_5pt_M1e6_k100","graph_M1e7_k200"):
    d=json.load(open("gpurun_out/r5e/bench_%s.json"%w)); p=d["partial_reorth"]
    print(w, d["value"], d["roofline"]["kernel"], d["roofline"].get("traffic_source","")[-60:], "partial", p["ms_per_solve"], "gram", d["ritz_gram"]["ms"], d["ritz_gram"]["frac"])
PY
