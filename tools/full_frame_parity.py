"""One-off validation: the WHOLE headline frame (1920x1080, 500 spp, 50 bounces) on the GPU against
the CPU oracle, every pixel, bit for bit.  ~90 s of 16 host threads."""
import json, os, sys, time, hashlib
sys.path.insert(0,'tests'); sys.path.insert(0,'ray-tracing-practice_amd')
import oracle_bindings as ob, rtp_bindings as rb, numpy as np
if os.environ.get('SCENE')=='default':      # the reference's `main --default` animation, frame FRAME
    host=rb.HostScene.from_config(rb.host_lib().rtp_host_default_config().decode())
    cam=host.frame_camera(int(os.environ.get('FRAME',0)))
    W,H,SPP=cam.image_width,cam.image_height,cam.samples_per_pixel
    label=f'reference default config frame {os.environ.get("FRAME",0)}: polyhedra scene {W}x{H} {SPP} spp depth {cam.max_depth}'
elif os.environ.get('SCENE')=='c5':         # BASELINE configs[4]: ~100k spheres + textured quad at 4K, SPP samples (default 16)
    W,H,SPP=3840,2160,int(os.environ.get('SPP',16))
    host=rb.HostScene.rtiow(half_extent=158, textured_quad=True, texture_size=2048)
    cam=rb.rtiow_camera(W,H,SPP,50)
    label=f'S-100k (99857 spheres + textured quad) {W}x{H} {SPP} spp 50 bounces'
else:
    W,H,SPP=1920,1080,int(os.environ.get('SPP',500))
    host=rb.HostScene.rtiow()
    view=os.environ.get('VIEW','headline')
    if view=='low':      # skims the ground under the spheres: rays that graze sphere bottoms and box faces
        cam=rb.make_camera(W,H,35.0,(-12.0,0.6,0.12),(4.0,0.0,0.2),(0.7,0.8,1.0),SPP,50)
    elif view=='top':    # straight down from far above: every primary ray is a far-origin ray for the cluster
        cam=rb.make_camera(W,H,12.0,(0.5,0.25,140.0),(0.0,0.0,0.0),(0.7,0.8,1.0),SPP,50)
    else:
        cam=rb.rtiow_camera(W,H,SPP,50)
    label=f'S-rtiow {W}x{H} {SPP} spp 50 bounces, view {view}'
dev=rb.DeviceScene(host,device=0)
got,tm=dev.render_to_host(cam)
print('gpu frame', tm.kernel_ms,'ms', flush=True)
t=time.time()
bad=0; maxabs=0.0
for r0 in range(0,H,60):
    want=ob.render(host,cam,row0=r0,row1=min(H,r0+60),threads=16)
    g=got[r0:r0+60]
    eq=(want.view(np.uint32)==g.view(np.uint32)).all(-1)
    bad+=int((~eq).sum()); maxabs=max(maxabs,float(np.abs(want-g).max()))
    print('rows',r0,'bad so far',bad,'elapsed %.0fs'%(time.time()-t), flush=True)
res={'config':label,'pixels':W*H,'samples':W*H*SPP,'pixels_differing':bad,'max_abs_diff':maxabs,
     'gpu_kernel_ms':tm.kernel_ms,'guarded_walk':int(tm.guarded),'guard_unproven':int(tm.guard_unproven),'trace_launches':int(tm.trace_launches),'primary_visibility':int(tm.primary_visibility),'sphere_only':int(tm.sphere_only),'guard_dynamic':int(tm.guard_dynamic),'flagged_samples':int(tm.flagged_samples),'oracle_seconds':time.time()-t,'frame_sha256':hashlib.sha256(got.tobytes()).hexdigest()}
print(json.dumps(res))
json.dump(res,open('gpurun_out/full_frame_parity_%s%s.json'%(os.environ.get('SCENE','rtiow'),'' if os.environ.get('VIEW','headline')=='headline' else '_'+os.environ['VIEW']),'w'),indent=1)
