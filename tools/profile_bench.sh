#!/bin/bash
# Profiles of the default bench.py run, as docs/LOG.md §6 / profiles/rNN cite them.  Run on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/profile_bench.sh r01'
# Kernel trace and each PMC group are separate rocprofv3 runs (never --pmc together with tracing).
set -eo pipefail
ROUND=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# under rocprofv3 bench.py runs as ONE RANK (WORLD_SIZE set), not as the launcher that starts child processes
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
BENCH="python3 $ROOT/bench.py --gpus 1 --steps 4 --warmup 0 --no-cpu-baseline"
ONE="python3 $ROOT/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace done"
pmc() {  # name, counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- $ONE > "$OUT/$name.log" 2>&1
    echo "$name done"
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc valu SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES
pmc wait GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES
unset WORLD_SIZE RANK LOCAL_RANK
python3 $ROOT/bench.py --steps 5 --warmup 2 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"      # the launcher: with its own live PMC passes
find "$OUT" -name '*.csv' | head -40
