"""Dev tool (GPU): 4-wide nodes vs child-pair nodes — S-rtiow 1920x1080 (SPP, default 100) and S-100k 4K (C5SPP, default 16);
frames compared with each other, times from rt_timing."""
import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'ray-tracing-practice_amd')
import numpy as np
import rtp_bindings as rb
rb.HONOUR_ENV = True      # developer tool: RTP_* variables steer the handles made below
SPP = int(os.environ.get('SPP', 100)); C5SPP = int(os.environ.get('C5SPP', 16))
for label, host, cam in (('S-rtiow 1080p', rb.HostScene.rtiow(), rb.rtiow_camera(1920, 1080, SPP, 50)),
                         ('S-100k 4K', rb.HostScene.rtiow(half_extent=158, textured_quad=True, texture_size=1024), rb.rtiow_camera(3840, 2160, C5SPP, 50))):
    ref = None
    for wide in (1, 0):
        dev = rb.DeviceScene(host, device=0, wide_nodes=wide)
        best = 1e9
        for _ in range(3):
            fb, t = dev.render_to_host(cam)
            best = min(best, t.trace_ms)
        if ref is None:
            ref = fb
        n = cam.image_width * cam.image_height * cam.samples_per_pixel
        print(f'{label} wide={t.wide_nodes} dyn={t.guard_dynamic} in_lds={t.scene_in_lds} lds={t.lds_bytes}: trace {best:.2f} ms = {n / best / 1e3:.1f} Msamples/s, '
              f'flagged {t.flagged_samples} ({100.0 * t.flagged_samples / n:.4f} %), same frame as first: {bool(np.array_equal(ref.view(np.uint32), fb.view(np.uint32)))}', flush=True)
        dev.close()
