"""Dev tool (GPU): sha256 of the whole frames of profiles/<round>/full_frame_parity*.json rendered with the library as it is now,
against the hashes those files hold (frames that were compared with the oracle pixel by pixel when the files were made) —
seconds instead of the minutes of oracle time a full comparison takes.  Usage: python tools/frame_hashes.py r03"""
import glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "ray-tracing-practice_amd"))
import rtp_bindings as rb

rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
W, H = 1920, 1080
sky = (0.7, 0.8, 1.0)


def setup(name):
    if name.endswith("_default.json"):
        host = rb.HostScene.from_config(rb.host_lib().rtp_host_default_config().decode())
        return host, host.frame_camera(7)
    if name.endswith("_c5.json"):
        return rb.HostScene.rtiow(half_extent=158, textured_quad=True, texture_size=2048), rb.rtiow_camera(3840, 2160, 16, 50)
    host = rb.HostScene.rtiow()
    if name.endswith("_low.json"):
        return host, rb.make_camera(W, H, 35.0, (-12.0, 0.6, 0.12), (4.0, 0.0, 0.2), sky, 500, 50)
    if name.endswith("_top.json"):
        return host, rb.make_camera(W, H, 12.0, (0.5, 0.25, 140.0), (0.0, 0.0, 0.0), sky, 500, 50)
    return host, rb.rtiow_camera(W, H, 500, 50)


out = []
for path in sorted(glob.glob(os.path.join(ROOT, "profiles", rnd, "full_frame_parity*.json"))):
    ref = json.load(open(path))
    host, cam = setup(os.path.basename(path))
    if cam.image_width * cam.image_height == ref["pixels"]:          # (the extra views were made at fewer samples per pixel)
        cam.samples_per_pixel = ref["samples"] // ref["pixels"]
    assert cam.image_width * cam.image_height == ref["pixels"] and cam.image_width * cam.image_height * cam.samples_per_pixel == ref["samples"], path
    dev = rb.DeviceScene(host, device=0)
    fb, tm = dev.render_to_host(cam)
    sha = hashlib.sha256(fb.tobytes()).hexdigest()
    out.append({"file": os.path.basename(path), "config": ref["config"], "sha256_now": sha, "same_as_file": sha == ref["frame_sha256"],
                "gpu_kernel_ms": round(tm.kernel_ms, 2), "primary_visibility": int(tm.primary_visibility), "flagged_samples": int(tm.flagged_samples)})
    print(json.dumps(out[-1]), flush=True)
    dev.close()
print(json.dumps({"version": rb.amd_lib().rt_version_string().decode(), "all_same": all(o["same_as_file"] for o in out), "frames": out}))
