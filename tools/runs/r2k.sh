#!/bin/bash
# round 2, GPU call K: fused-launch small path (tests + C1 bench), then PMC HBM-traffic passes of every BASELINE config
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_small.py tests/test_gpu_lanczos.py -m gpu -q -x -s > $O/pytest_small.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "fused-launch|small-engine|passed|failed|Error" $O/pytest_small.log | head -40
[ $rc -eq 0 ] || { grep -E "^E" $O/pytest_small.log | head -20; exit $rc; }
for w in dense_M512_k20 lap2d_5pt_M1e6_k100; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-partial > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-partial --no-cpu-baseline --tune 15=1 > $O/bench_${w}_plain.json 2> $O/bench_${w}_plain.err
done
cp $ROOT/profiles/hbm_traffic.json $O/hbm_traffic.json
for w in graph_M1e7_k200 lap2d_5pt_M1e7_k200 lap2d_5pt_M1e7_k500 lap3d_7pt_M1e8_k200 lap2d_5pt_M1e6_k100; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${w}_$c -o p -- python3 $ROOT/bench.py --workload $w --steps 1 --warmup 0 --no-prewarm --no-partial --no-cpu-baseline --no-profile > $O/pmc_${w}_$c.out 2> $O/pmc_${w}_$c.err); echo "pmc $w $c rc=$?"
  done
  python3 tools/make_traffic.py $w $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE $O/hbm_traffic.json > $O/traffic_$w.txt 2>&1; tail -25 $O/traffic_$w.txt
  rm -rf $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2k"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["roofline_all"].items()})
PY
