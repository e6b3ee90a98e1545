"""Developer tool (GPU): flagged fraction of the guarded walk on the reference's default scene."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-practice_amd"))
import rtp_bindings as rb
rb.HONOUR_ENV = True      # developer tool: RTP_* variables steer the handles made below
host = rb.HostScene.from_config(rb.host_lib().rtp_host_default_config().decode())
cam = host.frame_camera(7)
cam.samples_per_pixel = int(os.environ.get("SPP", "100"))
dev = rb.DeviceScene(host, 0)
n = cam.image_width * cam.image_height * cam.samples_per_pixel
print("spheres", host.desc.num_spheres, "planes", host.desc.num_planes, "nodes", host.desc.num_nodes, "reason", repr(dev.guard_reason()))
for env in ({}, {"RTP_TRAVERSAL": "guarded"}, {"RTP_TRAVERSAL": "threaded"}):
    for k in ("RTP_WGS_PER_CU", "RTP_STACK_LEVELS", "RTP_TRAVERSAL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    fb, t = dev.render_to_host(cam)
    fb, t = dev.render_to_host(cam)
    if os.environ.get("STATS"):
        import ctypes as C
        out = (C.c_uint32 * 16)()
        rb.amd_lib().rt_debug_read_stats(dev._h, out)
        print("   flag reasons (stack, tie, final; the rest: far origin):", out[12], out[13], out[14], "of", t.flagged_samples)
    print(env, f"kernel {t.kernel_ms:.1f} ms {n / t.kernel_ms / 1e3:.0f} Ms/s guarded {t.guarded} flagged {100.0 * t.flagged_samples / n:.3f} % lds {t.lds_bytes} wgs {t.num_workgroups}", flush=True)
