"""Experiment (GPU): one frame rendered as K row shards on K streams / K scene handles concurrently — the experiment
that exposed the contended work counter (docs/LOG.md §6).  One script for the three sweeps that used to be separate files:

  KS=1,2,4 python tools/multistream_exp.py                 K concurrent shard launches, frames checked against K=1
  PRIOS=1 python tools/multistream_exp.py                  K=2 with equal / different stream priorities
  WGS=1,0 KS=2,3,4,6,8 python tools/multistream_exp.py     sweep over workgroups per CU (0 = the default)
Environment: SPP (500), H (1080)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-practice_amd"))
import numpy as np
import torch
import frame_parallel as fp
import rtp_bindings as rb
rb.HONOUR_ENV = True      # developer tool: RTP_* variables steer the handles made below

host = rb.HostScene.rtiow()
SPP = int(os.environ.get("SPP", "500"))
H = int(os.environ.get("H", "1080"))
cam = rb.rtiow_camera(1920, H, SPP, 50)
ref = None


def run(K, wgs=0, prios=None, label=""):
    global ref
    devs = [rb.DeviceScene(host, 0, workgroups_per_cu=wgs) for _ in range(K)]
    streams = [torch.cuda.Stream(priority=p) for p in (prios or [0] * K)]
    shards = [rb.Shard(8, K, r) if K > 1 else None for r in range(K)]
    rows = [rb.amd_lib().rt_shard_rows(H, ctypes.byref(s) if s else None) for s in shards]
    fbs = [torch.zeros((rows[r], 1920, 3), dtype=torch.float32, device="cuda:0") for r in range(K)]

    def frame():
        for r in range(K):
            devs[r].render(cam, fbs[r].data_ptr(), shard=shards[r], stream=streams[r].cuda_stream, sync=False)
        torch.cuda.synchronize()
    frame(); frame()
    t0 = time.perf_counter()
    for _ in range(3):
        frame()
    dt = (time.perf_counter() - t0) / 3
    full = np.zeros((H, 1920, 3), np.float32)
    for r in range(K):
        full[fp.shard_row_indices(H, 8, K, r) if K > 1 else slice(None)] = fbs[r].cpu().numpy()
    if ref is None:
        ref = full
    same = bool(np.array_equal(ref.view(np.uint32), full.view(np.uint32)))
    ts = [d.last_timing() for d in devs]
    print(f"H={H} K={K} wgs/CU={wgs or 'auto'} {label}: {dt * 1e3:7.2f} ms = {1920 * H * SPP / dt / 1e6:7.1f} Ms/s; per-call trace ms "
          + " ".join(f"{t.trace_ms:.1f}" for t in ts) + f"; frame identical to the first run: {same}", flush=True)
    del devs, fbs
    torch.cuda.empty_cache()


run(1)
if os.environ.get("PRIOS"):
    run(2, prios=[0, 0], label="same priority")
    run(2, prios=[0, -1], label="priorities 0,-1")
    run(2, wgs=1, prios=[0, -1], label="priorities 0,-1")
for K in [int(x) for x in os.environ.get("KS", "2,4").split(",") if x]:
    for wgs in [int(x) for x in os.environ.get("WGS", "0").split(",") if x]:
        run(K, wgs)
