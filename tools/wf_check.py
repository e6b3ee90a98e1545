"""Dev tool (GPU): the wavefront kernel (rt_config.kernel = RT_KERNEL_WAVEFRONT) against the default kernel and the
oracle on a small frame, then its time on a big one.  Env: W H SPP (big frame), PATHS, EXCH, STATS=1 (RTP_STATS build)."""
import ctypes as C
import os
import sys

sys.path.insert(0, 'tests'); sys.path.insert(0, 'ray-tracing-practice_amd')
import numpy as np
import rtp_bindings as rb
rb.HONOUR_ENV = True      # developer tool: RTP_* variables steer the handles made below
import oracle_bindings as ob

paths = int(os.environ.get('PATHS', 0)); exch = int(os.environ.get('EXCH', 0))
hs = rb.HostScene.rtiow()
small = rb.rtiow_camera(160, 90, 9, 50)
wf = rb.DeviceScene(hs, device=0, kernel=rb.KERNEL_WAVEFRONT, wavefront_paths=paths, wavefront_exchange=exch)
mega = rb.DeviceScene(hs, device=0, kernel=rb.KERNEL_MEGA)
a, ta = wf.render_to_host(small)
b, tb = mega.render_to_host(small)
want = ob.render(hs, small, threads=8)
print('small frame: wavefront kernel id', ta.kernel, 'guarded', ta.guarded, 'flagged', ta.flagged_samples, '| mega flagged', tb.flagged_samples)
print('  wavefront == oracle:', bool(np.array_equal(a.view(np.uint32), want.view(np.uint32))),
      ' mega == oracle:', bool(np.array_equal(b.view(np.uint32), want.view(np.uint32))), flush=True)
if not np.array_equal(a.view(np.uint32), want.view(np.uint32)):
    bad = np.argwhere((a.view(np.uint32) != want.view(np.uint32)).any(axis=2))
    print('  differing pixels', len(bad), bad[:5].tolist(), 'max abs', float(np.abs(a - want).max()))
    sys.exit(1)
W, H, SPP = int(os.environ.get('W', 1920)), int(os.environ.get('H', 1080)), int(os.environ.get('SPP', 64))
cam = rb.rtiow_camera(W, H, SPP, 50)
for name, ds in (('wavefront', wf), ('mega', mega)):
    best = 1e9
    for it in range(3):
        fb, tm = ds.render_to_host(cam)
        best = min(best, tm.trace_ms)
    print(f'{name:10s} {W}x{H}x{SPP}: trace {best:.3f} ms = {W * H * SPP / best / 1e3:.1f} Msamples/s  wgs {tm.num_workgroups} x {tm.workgroup_size}  lds {tm.lds_bytes}'
          f'  flagged {tm.flagged_samples}  sum {float(fb.sum()):.6e}', flush=True)
    if os.environ.get('STATS') and name == 'wavefront':
        out = (C.c_uint32 * 16)()
        rb.amd_lib().rt_debug_read_stats(ds._h, out)
        ns = W * H * SPP
        for k, n in enumerate(['pair', 'leaf', 'shade', 'generate', 'exchange']):
            it, ln = out[2 * k], out[2 * k + 1]
            print(f'  {n:9s} wave-steps {it:11d}  lane occupancy {ln / max(it, 1):.3f}  lane-steps per sample {ln * 64 / ns:.2f}')
