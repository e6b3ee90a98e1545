import sys, time, os
sys.path.insert(0,'tests'); sys.path.insert(0,'ray-tracing-practice_amd')
import oracle_bindings as ob, rtp_bindings as rb, numpy as np
print(rb.amd_lib().rt_version_string())
def check(name, hs, cam, nprobe=4000):
    ds = rb.DeviceScene(hs)
    rng = np.random.default_rng(1)
    ijs = np.stack([rng.integers(0,cam.image_width,nprobe), rng.integers(0,cam.image_height,nprobe), rng.integers(0,max(cam.samples_per_pixel,1),nprobe)],1).astype(np.int32)
    t=time.time(); rad, rays, seeds = ds.trace_samples(cam, ijs); tg=time.time()-t
    orad, orays, oseeds = ob.trace_samples(hs, cam, ijs)
    same = (rad.view(np.uint32)==orad.view(np.uint32)).all(1) & (rays==orays) & (seeds==oseeds)
    print(name,'probe: bit-identical samples', same.sum(),'/',nprobe, 'rays equal', (rays==orays).sum(), 'seed equal',(seeds==oseeds).sum(), 'maxabs', np.abs(rad-orad).max())
    bad=np.where(~same)[0][:5]
    for b in bad: print('  bad', ijs[b], rad[b], orad[b], rays[b], orays[b], seeds[b], oseeds[b])
    t=time.time(); fb, tm = ds.render_to_host(cam); tg=time.time()-t
    ofb = ob.render(hs, cam, threads=8)
    eq = (fb.view(np.uint32)==ofb.view(np.uint32)).all(2)
    ns = cam.image_width*cam.image_height*cam.samples_per_pixel
    print(name,'render: identical pixels', eq.sum(),'/',eq.size,'max abs diff',np.abs(fb-ofb).max(),'kernel ms',tm.kernel_ms,'Msamples/s',ns/tm.kernel_ms/1e3,'lds',tm.lds_bytes,'in_lds',tm.scene_in_lds,'wgs',tm.num_workgroups)
    return ds
cfg=open('tests/golden/test_config.txt').read()
hs=rb.HostScene.from_config(cfg); cam=hs.frame_camera(0)
check('testcfg',hs,cam)
hs2=rb.HostScene.rtiow(); cam2=rb.rtiow_camera(480,320,8,50)
check('rtiow',hs2,cam2)
cam3=rb.rtiow_camera(1920,1080,16,50)
ds=rb.DeviceScene(hs2)
for it in range(3):
    fb,tm=ds.render_to_host(cam3); print('rtiow 1080p 16spp kernel ms',tm.kernel_ms,'Msamples/s',1920*1080*16/tm.kernel_ms/1e3)
