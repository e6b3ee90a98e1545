"""Dev tool: kernel time of the rtiow frame under the env knobs given on the command line."""
import os, sys, time
_R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(_R,'tests')); sys.path.insert(0,os.path.join(_R,'ray-tracing-practice_amd'))
import rtp_bindings as rb, numpy as np
rb.HONOUR_ENV = True      # developer tool: RTP_* variables steer the handles made below
W,H,SPP=int(os.environ.get('W',1920)),int(os.environ.get('H',1080)),int(os.environ.get('SPP',16))
hs=rb.HostScene.rtiow(half_extent=int(os.environ.get('EXT',11)))
cam=rb.rtiow_camera(W,H,SPP,50)
ds=rb.DeviceScene(hs,device=0)
best=1e9
for it in range(int(os.environ.get('ITERS',4))):
    fb,tm=ds.render_to_host(cam); best=min(best,tm.kernel_ms)
print('cfg', {k:v for k,v in os.environ.items() if k.startswith('RTP_')}, 'best kernel ms %.3f'%best, 'Msamples/s %.1f'%(W*H*SPP/best/1e3), 'lds',tm.lds_bytes,'wgs',tm.num_workgroups, 'sum', float(fb.sum()),
      'trace ms %.3f primary ms %.3f (on %d) rework ms %.3f flagged %d' % (tm.trace_ms, tm.primary_ms, tm.primary_visibility, tm.rework_ms, tm.flagged_samples))

import ctypes as C
lib=rb.amd_lib()
if os.environ.get('STATS'):
    out=(C.c_uint32*16)()
    lib.rt_debug_read_stats(ds._h, out)
    print('  shade step split (kticks): shade() %d  store+next sample %d  begin_ray+guard (in shadephase) ' % (out[12], out[13]))
    names=['inner','leaf','shadephase','shade']
    ns=W*H*SPP
    for k,n in enumerate(names):
        it,ln=out[2*k],out[2*k+1]
        print('  %-10s kticks %9d'%(n,out[8+k]) if k<4 else '', end=' ')
        print(' wave-steps %10d  full-wave-equiv %10d  util %.3f  per-sample lane-steps %.2f'%(it,ln,ln/max(it,1), ln*64/ns))

if os.environ.get('QSTATS'):
    out=(C.c_uint32*12)()
    lib.rt_debug_read_stats(ds._h, out)
    print('  T: iters %d exchanges %d popped %d idle %d boxvotes %d boxlanes/64 %d'%tuple(out[0:6]))
    print('  S: batches %d - popped %d idle %d'%(out[6],out[8],out[9]))
