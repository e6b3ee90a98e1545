#!/bin/bash
# Dev tool (GPU box): headline frame time without counters or CPU baseline; prints value, ms/step, trace launch ms, re-walk ms.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $ROOT/bench.py --steps ${STEPS:-3} --warmup 1 --no-pmc --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value', d['value'], 'ms/step', d['ms_per_step'], 'trace launch', r['launch_ms'], 'x', r['launches_per_step'], 'rework', r['rework_launch_ms'], 'flagged', r['flagged_sample_fraction'])"
