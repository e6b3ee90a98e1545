"""Experiment (GPU): whole frames alternating between TWO scene handles on two streams (frame k on handle k % 2), so that the
tail of one frame (exact re-walk, accumulation) and the head of the next (candidate lists, primary pass) can overlap.
Environment: SPP (500), FRAMES (8)."""
import os
import sys
import time

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "ray-tracing-practice_amd"))
import numpy as np
import torch
import rtp_bindings as rb
rb.HONOUR_ENV = True      # developer tool: RTP_* variables steer the handles made below

host = rb.HostScene.rtiow()
SPP = int(os.environ.get("SPP", "500"))
N = int(os.environ.get("FRAMES", "8"))
cam = rb.rtiow_camera(1920, 1080, SPP, 50)


def run(K, prios=None):
    devs = [rb.DeviceScene(host, 0) for _ in range(K)]
    streams = [torch.cuda.Stream(priority=p) for p in (prios or [0] * K)]
    fbs = [torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0") for _ in range(K)]

    def frames(n):
        for k in range(n):
            devs[k % K].render(cam, fbs[k % K].data_ptr(), stream=streams[k % K].cuda_stream, sync=False)
        torch.cuda.synchronize()
    frames(2 * K)
    t0 = time.perf_counter()
    frames(N)
    dt = (time.perf_counter() - t0) / N
    same = all(bool(torch.equal(fbs[0], f)) for f in fbs)
    print(f"{K} handle(s) on {K} stream(s), priorities {prios or [0] * K}: {dt * 1e3:7.2f} ms per frame = {1920 * 1080 * SPP / dt / 1e6:8.1f} Msamples/s; frames equal: {same}", flush=True)
    return fbs[0].cpu().numpy()


a = run(1)
b = run(2)
c = run(2, prios=[0, -1])
print("same bits as one handle:", bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))), bool(np.array_equal(a.view(np.uint32), c.view(np.uint32))))
