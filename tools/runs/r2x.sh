#!/bin/bash
# round 2, GPU call X: the five-launch loop (three-term recurrence folded into pass 1): full suite, headline bench both ways
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2x; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 850 python -m pytest tests -m gpu -q -x -s --durations=8 > $O/pytest_gpu_full_suite.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu_full_suite.log; grep "three-term-fused\]" $O/pytest_gpu_full_suite.log
[ $rc -eq 0 ] || { grep -E "^E" $O/pytest_gpu_full_suite.log | head -20; exit $rc; }
timeout -k 10 300 python bench.py --no-partial > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout -k 10 300 python bench.py --no-partial --no-cpu-baseline --tune 15=1 > $O/bench_six_launch_loop.json 2> $O/bench_six_launch_loop.err; echo "bench six-launch rc=$?"
timeout -k 10 300 python bench.py --no-partial --no-cpu-baseline --workload lap2d_5pt_M1e6_k100 > $O/bench_lap2d_5pt_M1e6_k100.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"
timeout -k 10 300 python bench.py --no-partial --no-cpu-baseline --workload lap2d_5pt_M1e6_k100 --tune 15=1 > $O/bench_lap2d_5pt_M1e6_k100_six_launch_loop.json 2> $O/bench_c2b.err; echo "bench c2 six rc=$?"
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2x"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["ms_per_step"], {k:(v["avg_us"],v["frac"]) for k,v in d["roofline_all"].items()}, d["roofline"]["kernel"], d["roofline"]["frac"])
PY
