"""Dev probe (GPU): forced distance-aware margins on scenes of tools/guard_stress.py — LDS-resident kernel against the global-memory one."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-practice_amd"))
import rtp_bindings as rb, guard_stress
for seed, trial in [(4, 25), (5, 14), (4, 15), (1, 12), (1, 39), (2, 15), (5, 8), (5, 5), (6, 37), (3, 30)]:
    for k, sph, pl, mats, cam, spread in guard_stress.scenes(seed, trial + 1):
        if k != trial: continue
        host = rb.HostScene.from_arrays(sph, pl, mats)
        for what, kw in (("lds", {}), ("global", dict(scene_in_lds=0)), ("exact", dict(traversal=rb.TRAVERSAL_EXACT))):
            kw = dict(dict(traversal=rb.TRAVERSAL_GUARDED, guard_keep=1, guard_dynamic_margins=2), **kw)
            d = rb.DeviceScene(host, 0, **kw)
            d.render_to_host(cam)
            _, t = d.render_to_host(cam)
            print(f"seed {seed} scene {trial} {sph.shape[0]} spheres {pl.shape[0]} planes {what:7s}: {t.kernel_ms:7.2f} ms trace {t.trace_ms:6.2f} re-walk {t.rework_ms:6.2f} flagged {t.flagged_samples} lds {t.lds_bytes} in_lds {t.scene_in_lds} wgs {t.num_workgroups} dyn {t.guard_dynamic} prim {t.primary_visibility}", flush=True)
            d.close()
