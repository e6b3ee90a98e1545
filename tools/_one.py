import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-practice_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rtp_bindings as rb, guard_stress
seed, trial = int(os.environ.get("SEED", 5)), int(os.environ.get("TRIAL", 22))
for k, sph, pl, mats, cam, spread in guard_stress.scenes(seed, trial + 1, 48):
    pass
host = rb.HostScene.from_arrays(sph, pl, mats)
kw = eval(os.environ.get("KW", "{}"))
d = rb.DeviceScene(host, 0, **kw)
for k in range(3):
    _, t = d.render_to_host(cam)
    print(kw, f"frame {k}: {t.kernel_ms:7.2f} ms guarded {t.guarded} abandoned {t.abandoned_passes} paused {t.guard_paused} lds {t.lds_bytes} trace {t.trace_ms:6.2f} re-walk {t.rework_ms:6.2f} flagged {t.flagged_samples}", flush=True)
