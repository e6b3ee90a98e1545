#!/bin/bash
# Dev tool (GPU box): instruction mix of the dominant kernel of one bench frame (two counter passes, no tracing).
#   gpurun -- 'bash tools/pmc_mix.sh [spp] [tag]'
set -eo pipefail
SPP=${1:-100}; TAG=${2:-mix}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
ONE="python3 $ROOT/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --spp $SPP"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT \
   --output-format csv -d "$OUT/a" -o pmc -- $ONE > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS \
   --output-format csv -d "$OUT/b" -o pmc -- $ONE > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name']
        args = [a.strip() for a in kn.split('<', 1)[1].split('>', 1)[0].split(',')] if 'render_kernel<' in kn else []
        if (len(args) > 1 and args[1] == 'false') or '_wf' in kn:      # the guarded trace kernel (kThreaded = false), not the exact re-walk
            acc[r['Counter_Name']] += float(r['Counter_Value'])
tot = acc.get('SQ_INSTS_VALU', 1.0)
for k in sorted(acc):
    print(f'{k:28s} {acc[k]:.4g}  {acc[k] / tot:.3f} of VALU')
PY
