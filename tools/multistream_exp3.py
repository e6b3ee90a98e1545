"""Experiment (GPU): K concurrent row-shard launches per frame, sweep over K and workgroups per CU; frames checked."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-practice_amd"))
import numpy as np, torch
import rtp_bindings as rb
import frame_parallel as fp
host = rb.HostScene.rtiow()
H = int(os.environ.get("H", "1080"))
cam = rb.rtiow_camera(1920, H, 500, 50)
ref = None
def run(K, wgs):
    global ref
    if wgs: os.environ["RTP_WGS_PER_CU"] = str(wgs)
    else: os.environ.pop("RTP_WGS_PER_CU", None)
    devs = [rb.DeviceScene(host, 0) for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    shards = [rb.Shard(8, K, r) if K > 1 else None for r in range(K)]
    rows = [rb.amd_lib().rt_shard_rows(H, ctypes.byref(s) if s else None) for s in shards]
    fbs = [torch.zeros((rows[r], 1920, 3), dtype=torch.float32, device="cuda:0") for r in range(K)]
    def frame():
        for r in range(K):
            devs[r].render(cam, fbs[r].data_ptr(), shard=shards[r], stream=streams[r].cuda_stream, sync=False)
        torch.cuda.synchronize()
    frame(); frame()
    t0 = time.perf_counter()
    for _ in range(3): frame()
    dt = (time.perf_counter() - t0) / 3
    full = np.zeros((H, 1920, 3), np.float32)
    for r in range(K):
        full[fp.shard_row_indices(H, 8, K, r) if K > 1 else slice(None)] = fbs[r].cpu().numpy()
    if ref is None: ref = full
    same = bool(np.array_equal(ref.view(np.uint32), full.view(np.uint32)))
    print(f"H={H} K={K} wgs/CU={wgs or 2}: {dt * 1e3:7.2f} ms = {1920 * H * 500 / dt / 1e6:7.1f} Ms/s  frame identical to K=1: {same}", flush=True)
    del devs, fbs; torch.cuda.empty_cache()
run(1, 0)
for K in [int(x) for x in os.environ.get("KS", "2,3,4,6,8").split(",") if x]:
    for wgs in (1, 0):
        run(K, wgs)
