#!/bin/bash
# round 2, final GPU call: full GPU suite and the bench lines of every BASELINE config with the final code
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2final; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 850 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu_full_suite.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu_full_suite.log
[ $rc -eq 0 ] || { grep -E "^E" $O/pytest_gpu_full_suite.log | head -20; exit $rc; }
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench driver-style rc=$?"
for w in dense_M512_k20 lap2d_5pt_M1e6_k100 graph_M1e7_k200 lap2d_5pt_M1e7_k500; do
  timeout -k 10 400 python bench.py --workload $w --steps 3 --warmup 1 --no-partial > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
timeout -k 10 400 python bench.py --workload lap3d_7pt_M1e8_k200 --steps 1 --warmup 1 --no-partial --no-cpu-baseline --no-prewarm > $O/bench_lap3d_7pt_M1e8_k200.json 2> $O/bench_lap3d_7pt_M1e8_k200.err; echo "bench c4 rc=$?"
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2final"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["ms_per_step"], {k:(v["avg_us"],v["frac"]) for k,v in d["roofline_all"].items()}, d["roofline"]["kernel"], d["roofline"]["frac"], (d.get("ritz_backtransform") or {}).get("ms"), (d.get("ritz_backtransform") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
PY
