set -e
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for cfg in "400 225 16" "1200 800 10" "1200 800 100" "1920 1080 500"; do set -- $cfg; W=$1 H=$2 SPP=$3 ITERS=4 python3 tools/perf_sweep.py | cut -c40-260; done
bash tools/_run.sh
cat > /tmp/sh2.py <<'PY'
import sys, os, ctypes
R=os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0,os.path.join(R,'tests')); sys.path.insert(0,os.path.join(R,'ray-tracing-practice_amd'))
import rtp_bindings as rb, torch
host=rb.HostScene.rtiow()
cam=rb.rtiow_camera(1920,1080,500,50)
ds=rb.DeviceScene(host,device=0)
base=None
for N in (1,2,4,8):
    for r in sorted({0,N-1}):
        sh=rb.Shard(8,N,r) if N>1 else None
        rows=rb.amd_lib().rt_shard_rows(1080, ctypes.byref(sh) if sh else None)
        fb=torch.zeros((rows,1920,3),dtype=torch.float32,device='cuda:0')
        best=1e9
        for i in range(4):
            ds.render(cam, fb.data_ptr(), shard=sh, sync=True)
            tm=ds.last_timing()
            if tm.kernel_ms<best: best=tm.kernel_ms; keep=(tm.trace_ms,tm.primary_ms,tm.rework_ms,tm.flagged_samples)
        if base is None: base=best
        print('N=%d rank %d rows %4d: kernel %.2f ms (ideal %.2f, efficiency %.3f) trace %.2f primary %.2f rework %.2f flagged %d'%(N,r,rows,best,base*rows/1080,base*rows/1080/best,*keep),flush=True)
PY
python3 /tmp/sh2.py
echo nosplit; RTP_AMD_LIB=ray-tracing-practice_amd/variants/librtp_amd_nosplit.so python3 /tmp/sh2.py
