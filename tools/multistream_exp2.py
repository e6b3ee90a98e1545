"""Experiment (GPU): two half-frame trace launches truly concurrent (different HW queues via stream priority)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-practice_amd"))
import torch
import rtp_bindings as rb
host = rb.HostScene.rtiow()
cam = rb.rtiow_camera(1920, 1080, 500, 50)
def run(K, prios, label):
    devs = [rb.DeviceScene(host, 0) for _ in range(K)]
    streams = [torch.cuda.Stream(priority=p) for p in prios]
    shards = [rb.Shard(8, K, r) if K > 1 else None for r in range(K)]
    rows = [rb.amd_lib().rt_shard_rows(1080, ctypes.byref(s) if s else None) for s in shards]
    fbs = [torch.zeros((rows[r], 1920, 3), dtype=torch.float32, device="cuda:0") for r in range(K)]
    def frame():
        for r in range(K):
            devs[r].render(cam, fbs[r].data_ptr(), shard=shards[r], stream=streams[r].cuda_stream, sync=False)
        torch.cuda.synchronize()
    frame(); frame()
    t0 = time.perf_counter()
    for _ in range(4): frame()
    dt = (time.perf_counter() - t0) / 4
    ts = [d.last_timing() for d in devs]
    print(f"{label}: {dt * 1e3:.2f} ms per frame = {1920 * 1080 * 500 / dt / 1e6:.1f} Msamples/s; per-call trace ms " + " ".join(f"{t.trace_ms:.1f}" for t in ts) + f"  wgs {ts[0].num_workgroups}", flush=True)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
run(1, [0], "K=1")
run(2, [0, 0], "K=2 same priority")
run(2, [0, -1], "K=2 priorities 0,-1")
os.environ["RTP_WGS_PER_CU"] = "1"
run(2, [0, -1], "K=2 priorities 0,-1, 1 WG/CU each")
run(1, [0], "K=1, 1 WG/CU")
os.environ["RTP_WGS_PER_CU"] = "1"
run(2, [0, 0], "K=2 same priority, 1 WG/CU each")
run(4, [0, 0, 0, 0], "K=4 same priority, 1 WG/CU each")
