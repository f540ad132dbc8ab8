#!/bin/bash
# round 2, GPU call L: what the driver runs at round end - whole -m gpu suite, smoke(), default bench - plus C1 / C2 lines
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2l; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -14 $O/pytest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench driver-style rc=$?"
for w in dense_M512_k20 lap2d_5pt_M1e6_k100; do timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-partial > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"; done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2l"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    r=d.get("roofline") or {}
    print(os.path.basename(f), d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("traffic"), {k:v["avg_us"] for k,v in d["roofline_all"].items()}, d["ritz_backtransform"]["ms"], d["ritz_backtransform"]["frac"], (d.get("cpu_baseline") or {}).get("value"))
PY
