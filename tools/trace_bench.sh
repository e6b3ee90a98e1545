#!/bin/bash
# Dev tool (GPU box): per-kernel times of bench frames (rocprofv3 --kernel-trace --stats; no counters in this run).
#   gpurun -- 'bash tools/trace_bench.sh TAG [extra bench args]'   → gpurun_out/trace_TAG/…kernel_stats.csv
set -eo pipefail
TAG=${1:-t}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 4 --warmup 0 --no-cpu-baseline "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
