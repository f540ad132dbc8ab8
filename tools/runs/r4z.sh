#!/bin/bash
# round 4, last GPU call: full GPU suite + smoke + driver-style and default bench lines on the final tree
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r4z; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu_full_suite.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu_full_suite.log
[ $rc -eq 0 ] || { grep -E "^E" $O/pytest_gpu_full_suite.log | head -20; exit $rc; }
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench driver-style rc=$?"
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python3 - <<'PY'
import json
for f in ("bench_driver_style","bench_default"):
    d=json.loads(open("gpurun_out/r4z/%s.json"%f).read().strip().splitlines()[-1])
    p=d["partial_reorth"]; c=d["class_surface"]
    print(f, d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic_source"][:90])
    print("  partial", p["ms_per_solve"], p["iterations_per_s"], p["host_syncs_inside_lz_run"], p["whole_iteration_frac_hbm_peak"])
    print("  ritz", d["ritz_backtransform"]["ms"], "gram", d["ritz_gram"]["ms"], d["ritz_gram"]["mfma_issue_utilisation_in_cycles"])
    print("  class", c["first_call"]["overhead_s"], c["second_call"]["overhead_s"], c["H_eigvals_s"], c["V_fetch_s"], c["H_eigvecs_fetch_s"])
PY
