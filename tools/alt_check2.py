import sys, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'ray-tracing-practice_amd')
import rtp_bindings as rb, numpy as np
os.environ['RTP_TRAVERSAL']='ordered'
host=rb.HostScene.rtiow(); dev=rb.DeviceScene(host,device=0)
w,h,spp=[int(x) for x in sys.argv[1:4]]
cam=rb.rtiow_camera(w,h,spp,50)
t=time.time(); got,tm=dev.render_to_host(cam); print('ordered',w,h,spp,'done',time.time()-t, tm.lds_bytes, tm.num_workgroups, flush=True)
