import sys, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'ray-tracing-practice_amd')
import rtp_bindings as rb, numpy as np
host=rb.HostScene.rtiow(); dev=rb.DeviceScene(host,device=0)
cam=rb.rtiow_camera(320,200,8,50)
want,_=dev.render_to_host(cam); print('default ok', flush=True)
mode=sys.argv[1]
if mode=='queue': os.environ['RTP_KERNEL']='queue'
else: os.environ['RTP_TRAVERSAL']='ordered'
t=time.time()
try:
    got,tm=dev.render_to_host(cam); print(mode,'done in',time.time()-t,'same',np.array_equal(got.view(np.uint32),want.view(np.uint32)), flush=True)
except Exception as e:
    print(mode,'error',e, flush=True)
