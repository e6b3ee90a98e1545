"""Developer tool (GPU): flagged-sample fraction and timing of the guarded walk on S-rtiow."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-practice_amd"))
import numpy as np
import rtp_bindings as rb
rb.HONOUR_ENV = True      # developer tool: RTP_* variables steer the handles made below

host = rb.HostScene.rtiow()
dev = rb.DeviceScene(host, 0)
W, H, SPP = 1920, 1080, int(os.environ.get("SPP", "128"))
cam = rb.rtiow_camera(W, H, SPP, 50)
for mode in ("guarded", "threaded"):
    os.environ["RTP_TRAVERSAL"] = mode
    fb, t = dev.render_to_host(cam)
    fb, t = dev.render_to_host(cam)
    n = W * H * SPP
    print(f"{mode:9s} kernel {t.kernel_ms:8.2f} ms  {n / t.kernel_ms / 1e3:8.1f} Msamples/s  guarded={t.guarded} flagged={t.flagged_samples} "
          f"({100.0 * t.flagged_samples / n:.4f} %)  wgs {t.num_workgroups} lds {t.lds_bytes}", flush=True)
    if mode == "guarded":
        ref = fb
    else:
        print("frames bit-identical:", bool(np.array_equal(ref.view(np.uint32), fb.view(np.uint32))))
