"""C5 (S-100k + textured quad, 3840x2160) one pass of SPP samples: time, and with STATS=1 the step counters of an RTP_STATS build."""
import os, sys, time
_R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(_R,'tests')); sys.path.insert(0,os.path.join(_R,'ray-tracing-practice_amd'))
import rtp_bindings as rb, numpy as np, ctypes as C
rb.HONOUR_ENV = True
import torch
spp=int(os.environ.get('SPP',125))
hs=rb.HostScene.rtiow(half_extent=158, textured_quad=True, texture_size=int(os.environ.get("TEXSIZE",2048)))
cam=rb.rtiow_camera(3840,2160,spp,50)
ds=rb.DeviceScene(hs,device=0, **eval(os.environ.get('KW','{}')))
fb=torch.zeros((2160,3840,3),dtype=torch.float32,device='cuda:0')
best=None
for it in range(int(os.environ.get('ITERS',2))):
    tm=ds.render(cam, fb.data_ptr())
    if best is None or tm.kernel_ms<best.kernel_ms: best=tm
tm=best
ns=3840*2160*spp
import hashlib
print('C5 %d spp: kernel_ms %.1f trace_ms %.1f primary %.1f rework %.1f launches %d  Msamples/s %.1f lds %d guarded %d dyn %d prim %d flagged %.4f %% vgprs %d scratch %d wgs %d x %d sha %s'%(spp,tm.kernel_ms,tm.trace_ms,tm.primary_ms,tm.rework_ms,tm.trace_launches,ns/tm.kernel_ms/1e3,tm.lds_bytes,tm.guarded,tm.guard_dynamic,tm.primary_visibility,100.0*tm.flagged_samples/ns,tm.trace_vgprs,tm.trace_scratch_bytes,tm.num_workgroups,tm.workgroup_size,hashlib.sha256(fb.cpu().numpy().tobytes()).hexdigest()[:12]))
if os.environ.get('STATS'):
    lib=rb.amd_lib(); out=(C.c_uint32*16)(); lib.rt_debug_read_stats(ds._h, out)
    print('  shade block split (kticks): (A) store+fetch %d   (C)+(B) material+arm %d   rest %d   votes %d' % (out[12], out[13], out[10], out[11]))
    print('  why flagged (RTP_STATS without the shade split): full stack %d, exact tie %d, final check / Schlick window %d' % (out[12], out[13], out[14]))
    names=['pair','leaf','shadephase','shade']
    for k,n in enumerate(names):
        it,ln=out[2*k],out[2*k+1]
        print('  %-10s kticks %9d wave-steps %10d full-wave-equiv %10d util %.3f per-sample lane-steps %.2f wave-steps/sample %.4f'%(n,out[8+k],it,ln,ln/max(it,1),ln*64/ns,it/ns))
