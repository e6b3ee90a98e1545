#!/bin/bash
# Dev tool (GPU box): per-kernel counter sums of ONE bench frame, every kernel of the frame (counter passes only, no tracing).
#   gpurun -- 'bash tools/pmc_kernels.sh [spp] [tag]'   → gpurun_out/pmck_TAG/summary.txt
set -eo pipefail
SPP=${1:-500}; TAG=${2:-k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmck_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
ONE="python3 $ROOT/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --spp $SPP"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
   --output-format csv -d "$OUT/a" -o pmc -- $ONE > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA \
   --output-format csv -d "$OUT/b" -o pmc -- $ONE > "$OUT/b.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/c" -o pmc -- $ONE > "$OUT/c.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/d" -o pmc -- $ONE > "$OUT/d.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:64]][r['Counter_Name']] += float(r['Counter_Value'])
for kn in sorted(acc):
    if 'rocclr' in kn or 'at::' in kn: continue
    print(kn)
    for k in sorted(acc[kn]): print(f'    {k:26s} {acc[kn][k]:.5g}')
PY
