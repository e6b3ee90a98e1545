"""Dev tool (GPU): the S-rtiow scene family over sizes (half_extent = 11 … 158: 486 … 100 k spheres) at 1920x1080 — which walk a default
handle takes and what it delivers; KW="dict(...)" = rt_config fields."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-practice_amd"))
import rtp_bindings as rb
spp = int(os.environ.get("SPP", 100))
kw = eval(os.environ.get("KW", "{}"))
for half in [int(x) for x in os.environ.get("HALVES", "11,14,18,22,32,48,64,100,158").split(",")]:
    host = rb.HostScene.rtiow(half_extent=half)
    cam = rb.rtiow_camera(1920, 1080, spp, 50)
    dev = rb.DeviceScene(host, 0, **kw)
    best = None
    for _ in range(3):
        _, t = dev.render_to_host(cam)
        if best is None or t.kernel_ms < best.kernel_ms: best = t
    t = best
    print(f"half_extent {half:3d}: {host.desc.num_spheres:6d} spheres  {1920 * 1080 * spp / t.kernel_ms / 1e3:8.0f} Msamples/s  kernel {t.kernel_ms:7.2f} ms trace {t.trace_ms:7.2f} primary {t.primary_ms:5.2f} "
          f"re-walk {t.rework_ms:5.2f}  guarded {t.guarded} in_lds {t.scene_in_lds} dyn {t.guard_dynamic} wide {t.wide_nodes} simple {t.sphere_only} prim {t.primary_visibility} "
          f"front {t.front_primitives} flagged {100.0 * t.flagged_samples / (1920 * 1080 * spp):.4f} % wgs {t.num_workgroups} x {t.workgroup_size} lds {t.lds_bytes} reason {dev.guard_reason()!r}", flush=True)
    dev.close()
