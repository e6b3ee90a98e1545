#!/bin/bash
# Dev tool (GPU box): VALU / wave-time counters of the kernels of ONE headline pass (tools/headline_pass.py) under rt_config fields
# given as KW="dict(...)" — e.g. KW="dict(traversal=1)" for the exact walk.   gpurun -- 'KW="dict(traversal=1)" bash tools/pmc_pass.sh TAG'
set -eo pipefail
TAG=${1:-p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcp_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp ITERS=1
RUN="python3 $ROOT/tools/headline_pass.py"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/a" -o pmc -- $RUN > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$OUT/b" -o pmc -- $RUN > "$OUT/b.log" 2>&1
grep HL "$OUT/a.log" || true
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:70]][r['Counter_Name']] += float(r['Counter_Value'])
for kn in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0)):
    m = acc[kn]
    if 'render_kernel' not in kn: continue
    cyc = m['GRBM_GUI_ACTIVE'] / 8
    print(kn)
    print('   VALU wave-instr %.4g  issue %.3f  lanes %.3f  | wave time: issuing %.3f waitcnt %.3f stalled %.3f | SALU/VALU %.2f LDS/VALU %.3f VMEM_RD %.4g VMEM_WR %.4g | LDS bank-conflict share %.3f' % (
        m['SQ_INSTS_VALU'], 2 * m['SQ_INSTS_VALU'] / (1024 * cyc), m['SQ_THREAD_CYCLES_VALU'] / (64 * m['SQ_ACTIVE_INST_VALU']),
        m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'],
        m['SQ_INSTS_SALU'] / m['SQ_INSTS_VALU'], m['SQ_INSTS_LDS'] / m['SQ_INSTS_VALU'], m['SQ_INSTS_VMEM_RD'], m['SQ_INSTS_VMEM_WR'],
        m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1)))
PY
