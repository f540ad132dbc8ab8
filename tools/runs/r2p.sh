#!/bin/bash
# round 2, GPU call P: tile padding of the two-phase SpMV: 8 (one 64-byte sector) vs 4 vs 2 products
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r2p; mkdir -p $O
export TMPDIR=/tmp
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$ROOT/include -I/opt/rocm/include -Wno-unused-result -Wno-unused-value"
for pad in 4 2; do
  (cd lanczos_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -DLZ_PB_PAD=$pad -c lz_spmv_pb.hip -o obj/lz_spmv_pb.o && make > /dev/null) || exit 1
  timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "two_phase or spmv" > $O/pytest_pad$pad.log 2>&1; rc=$?; echo "pad $pad pytest rc=$rc"; tail -1 $O/pytest_pad$pad.log
  [ $rc -eq 0 ] || { grep -E "^E" $O/pytest_pad$pad.log | head; exit $rc; }
  timeout -k 10 300 python bench.py --workload graph_M1e7_k200 --steps 2 --warmup 1 --no-partial --no-cpu-baseline > $O/bench_c3_pad$pad.json 2> $O/bench_c3_pad$pad.err; echo "bench c3 pad $pad rc=$?"
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("O","gpurun_out/r2p"),"bench_*.json"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(os.path.basename(f), d["value"], d["config"].get("spmv_kernel"), {k:v["avg_us"] for k,v in d["roofline_all"].items()})
PY
