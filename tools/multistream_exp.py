"""Experiment (GPU): one frame rendered as K row shards on K streams / K scene handles concurrently."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-practice_amd"))
import torch
import rtp_bindings as rb
host = rb.HostScene.rtiow()
SPP = int(os.environ.get("SPP", "500"))
cam = rb.rtiow_camera(1920, 1080, SPP, 50)
for K in [int(x) for x in os.environ.get("KS", "1,2,4").split(",")]:
    devs = [rb.DeviceScene(host, 0) for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    shards = [rb.Shard(8, K, r) if K > 1 else None for r in range(K)]
    rows = [rb.amd_lib().rt_shard_rows(1080, __import__("ctypes").byref(s) if s else None) for s in shards]
    fbs = [torch.zeros((rows[r], 1920, 3), dtype=torch.float32, device="cuda:0") for r in range(K)]
    def frame():
        for r in range(K):
            devs[r].render(cam, fbs[r].data_ptr(), shard=shards[r], stream=streams[r].cuda_stream, sync=False)
        torch.cuda.synchronize()
    frame(); frame()
    t0 = time.perf_counter()
    for _ in range(4):
        frame()
    dt = (time.perf_counter() - t0) / 4
    print(f"K={K}: {dt * 1e3:.2f} ms per frame = {1920 * 1080 * SPP / dt / 1e6:.1f} Msamples/s", flush=True)
    del devs, fbs
    torch.cuda.empty_cache()
