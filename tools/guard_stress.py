"""Developer tool (GPU): what a drop-in user gets by DEFAULT (rt_config as rt_config_init leaves it) against the exact walk
(itself checked against the oracle by the test suite) on many random scenes at scale — sphere scenes of 20 to 3000 spheres
with radii over three decades, with and without a huge ground sphere, mixed sphere/plane scenes, cameras inside, outside and
far away.  Per scene: a fresh handle's FIRST default frame (the one that pays for finding out that a scene does not suit the
guarded walk: in-launch bail-out), its second and third (after the handle's own judgement: step aside / one exact frame to time
against), the exact walk, and a frame with the guarded walk forced.  Reports differing pixels and the default ÷ exact ratios."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-practice_amd"))
import numpy as np
import rtp_bindings as rb

W, H = 1280, 720


def material(rng):
    m = rb.Material()
    m.type = int(rng.integers(0, 4)); m.fuzz = float(rng.uniform(0, 0.7)); m.ir = float(rng.uniform(1.1, 2.0))
    m.albedo.e[:] = rng.uniform(0.1, 1.0, 3)
    m.absorption.e[:] = rng.uniform(0, 0.6, 3) if rng.random() < 0.5 else (0, 0, 0)
    if os.environ.get("NOABS"):          # no absorbing glass: scenes without planes take the sphere-only builds (the draw above stays: same scenes otherwise)
        m.absorption.e[:] = (0, 0, 0)
    m.emit.e[:] = rng.uniform(0.5, 3.0, 3) if m.type == 3 else (0, 0, 0)
    m.texture_id = 0
    return m


def scenes(seed, count, spp=48, width=W, height=H):
    """The scenes of one seed, in order: (trial, spheres [n,5], planes [m,11], materials, camera, spread)."""
    rng = np.random.default_rng(seed)
    for trial in range(count):
        n = int(10 ** rng.uniform(1.3, 3.5))
        mats = [material(rng) for _ in range(8)]
        spread = float(rng.choice([2.0, 8.0, 30.0, 100.0]))
        sph = np.zeros((n, 5), np.float32)
        sph[:, :3] = rng.uniform(-spread, spread, (n, 3)) * np.array([1, 1, 0.3 if trial % 3 == 0 else 1.0])
        sph[:, 3] = 10.0 ** rng.uniform(-2.3, np.log10(spread) - 0.8, n)
        sph[:, 4] = rng.integers(0, len(mats), n)
        if trial % 2 == 0:
            sph[0] = (0, 0, -1000 - 0.3 * spread, 1000, 0)
        n_pl = int(rng.integers(0, 60)) if trial % 4 == 1 else 0
        pl = np.zeros((n_pl, 11), np.float32)
        if n_pl:
            pl[:, :3] = rng.uniform(-spread, spread, (n_pl, 3))
            pl[:, 3:6] = rng.uniform(-0.3 * spread, 0.3 * spread, (n_pl, 3))
            pl[:, 6:9] = rng.uniform(-0.3 * spread, 0.3 * spread, (n_pl, 3))
            pl[:, 9] = rng.integers(0, len(mats), n_pl)
            pl[:, 10] = rng.integers(0, 3, n_pl)
        eye = rng.uniform(-spread, spread, 3) * float(rng.choice([0.3, 1.0, 1.5, 6.0]))
        cam = rb.make_camera(width, height, float(rng.uniform(15, 90)), eye, rng.uniform(-0.3 * spread, 0.3 * spread, 3), rng.uniform(0, 1, 3), spp,
                             int(rng.integers(2, 50)))
        yield trial, sph, pl, mats, cam, spread


def main():
    seed, count, spp = int(os.environ.get("SEED", "1")), int(os.environ.get("SCENES", "24")), int(os.environ.get("SPP", "48"))
    total_bad = total_samples = over = 0
    worst = 0.0
    only = os.environ.get("TRIAL")          # one scene of the sequence, with the variants' timing details
    extra = eval(os.environ.get("EXTRA", "{}"))
    for trial, sph, pl, mats, cam, spread in scenes(seed, count, spp):
        if only is not None and trial != int(only):
            continue
        host = rb.HostScene.from_arrays(sph, pl, mats)
        if only is not None:
            for what, kw in (("default", {}), ("forced guarded, kept", dict(traversal=rb.TRAVERSAL_GUARDED, guard_keep=1)),
                             ("forced guarded", dict(traversal=rb.TRAVERSAL_GUARDED)), ("default, no bail-out", dict(guard_bail_share=-1)),
                             ("exact", dict(traversal=rb.TRAVERSAL_EXACT))):
                d = rb.DeviceScene(host, 0, **kw)
                for k in range(3):
                    _, t = d.render_to_host(cam)
                    print(f"{what:22s} frame {k}: {t.kernel_ms:7.2f} ms guarded {t.guarded} abandoned {t.abandoned_passes} paused {t.guard_paused} trace {t.trace_ms:6.2f} re-walk {t.rework_ms:6.2f} primary "
                          f"{t.primary_ms:5.2f} (on {t.primary_visibility}) flagged {t.flagged_samples} lds {t.lds_bytes} in_lds {t.scene_in_lds} wgs {t.num_workgroups} x {t.workgroup_size} "
                          f"dyn {t.guard_dynamic} simple {t.sphere_only} vgprs {t.trace_vgprs}", flush=True)
                    if os.environ.get("STATS") and k == 0:          # an RTP_STATS build (RTP_AMD_LIB): the frame's step counters, both launches together
                        import ctypes as C
                        out = (C.c_uint32 * 16)(); rb.amd_lib().rt_debug_read_stats(d._h, out)
                        ns = W * H * spp
                        print("    " + "  ".join(f"{n} wave-steps/sample {out[2 * i] / ns:.4f} lanes {out[2 * i + 1] / max(out[2 * i], 1):.3f} kticks {out[8 + i]}"
                                                 for i, n in enumerate(["pair", "leaf", "shade", "vote"])))
                        print(f"    why flagged: full stack {out[12]}, exact tie {out[13]}, final check / Schlick window {out[14]}", flush=True)
                d.close()
            continue
        n_samples = W * H * spp
        exact = rb.DeviceScene(host, 0, traversal=rb.TRAVERSAL_EXACT)
        exact.render_to_host(cam)                        # warm-up (code objects, clocks, the slab)
        e, te = exact.render_to_host(cam)
        exact.close()
        dev = rb.DeviceScene(host, 0, **extra)           # the defaults (EXTRA="dict(...)": rt_config fields to try on top of them)
        frames = [dev.render_to_host(cam) for _ in range(3)]
        dev.close()
        forced = rb.DeviceScene(host, 0, traversal=rb.TRAVERSAL_GUARDED, guard_keep=1, **extra)
        g, tg = forced.render_to_host(cam)
        reason = forced.guard_reason() or "eligible"
        forced.close()
        bad = sum(int((fb.view(np.uint32) != e.view(np.uint32)).any(axis=-1).sum()) for fb, _ in frames + [(g, tg)])
        total_bad += bad
        total_samples += n_samples
        ratios = [t.kernel_ms / te.kernel_ms for _, t in frames]
        worst = max(worst, *ratios)
        over += sum(int(r > 1.3) for r in ratios)
        t0 = frames[0][1]
        print(f"scene {trial:2d}: {sph.shape[0]:5d} spheres {pl.shape[0]:2d} planes spread {spread:5.1f} ({reason}) | default frames: "
              + ", ".join(f"{t.kernel_ms:7.2f} ms ({'guarded' if t.guarded else 'exact'}{', abandoned' if t.abandoned_passes else ''})" for _, t in frames)
              + f" | flagged {100.0 * t0.flagged_samples / n_samples:7.4f} % trace {t0.trace_ms:6.2f} re-walk {t0.rework_ms:6.2f} primary {t0.primary_ms:5.2f} | exact {te.kernel_ms:7.2f} ms"
              f" | forced guarded {tg.kernel_ms:7.2f} ms (trace {tg.trace_ms:.2f} re-walk {tg.rework_ms:.2f}), flagged {100.0 * tg.flagged_samples / n_samples:7.4f} % | ratios " + " ".join(f"{r:4.2f}" for r in ratios)
              + ("  <-- above 1.3" if max(ratios) > 1.3 else "") + f" | differing pixels {bad}", flush=True)
    print(f"TOTAL seed {seed}: {count} scenes, {total_samples / 1e9:.2f} G samples per frame set, differing pixels {total_bad}, worst default/exact ratio {worst:.2f}, "
          f"default frames above 1.3: {over}")


if __name__ == "__main__":
    main()
