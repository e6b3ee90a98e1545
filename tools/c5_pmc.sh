#!/bin/bash
# Dev tool (GPU box): PMC set of BASELINE configs[4] (S-100k + textured quad, 3840x2160) at SPP samples (default 64):
# VALU utilisation, wave-time split, L1 (TCP) and L2 (TCC) hit counters, HBM bytes — counters only, separate passes.
set -eo pipefail
SPP=${1:-64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c5_pmc_r04
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp SPP
RUN="python3 $ROOT/tools/c5_run.py"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/valu" -o pmc -- $RUN > "$OUT/valu.log" 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/cache" -o pmc -- $RUN > "$OUT/cache.log" 2>&1 || echo "cache counters unavailable"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o pmc -- $RUN > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o pmc -- $RUN > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'render_kernel<false, false' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
m = {k: acc[k] / cnt[k] for k in acc}
out = {'kernel': 'rtk::render_kernel<false, false, true, true, false, true> (guarded walk on 4-wide nodes fed by the primary-visibility pass, distance-aware margins in parametric form, records through L1/L2)', 'launches': cnt.get('SQ_INSTS_VALU', 0), 'per_launch_mean': m}
if 'GRBM_GUI_ACTIVE' in m:
    cycles = m['GRBM_GUI_ACTIVE'] / 8
    out['valu_issue_utilisation'] = round(2 * m['SQ_INSTS_VALU'] / (1024 * cycles), 4)
    out['valu_lane_utilisation'] = round(m['SQ_THREAD_CYCLES_VALU'] / (64 * m['SQ_ACTIVE_INST_VALU']), 4)
    out['wave_time'] = {k: round(m[c] / m['SQ_WAVE_CYCLES'], 3) for k, c in (('issuing', 'SQ_ACTIVE_INST_ANY'), ('s_waitcnt', 'SQ_WAIT_ANY'), ('issue_stalled', 'SQ_WAIT_INST_ANY'))}
if 'TCC_HIT_sum' in m:
    out['l2_hit_rate'] = round(m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']), 4)
if 'TCP_TOTAL_CACHE_ACCESSES_sum' in m and m['TCP_TOTAL_CACHE_ACCESSES_sum'] > 0:
    out['l1_hit_rate'] = round(1 - m['TCP_TCC_READ_REQ_sum'] / m['TCP_TOTAL_CACHE_ACCESSES_sum'], 4)
if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
    out['hbm_bytes_per_launch'] = int((2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024)
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[1] + '/c5_pmc.json', 'w'), indent=1)
PY
