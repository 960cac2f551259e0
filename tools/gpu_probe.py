"""First-contact GPU probe: smoke, traversal parity, quick timings (scratch tool, not part of the product)."""
import sys, time, os
sys.path.insert(0, "rust-pathtracer_amd"); sys.path.insert(0, "tests")
import numpy as np
import ptamd, orc
L = ptamd.lib()
print(L.pt_version())
t = time.time()
img = ptamd.render_rgb(64, 48, 4)
print("first render ok", time.time() - t, img.mean())
print(ptamd.scene_info())
rng = np.random.default_rng(1)
n = 100000
o = rng.uniform(-0.99, 0.99, (n, 3)).astype(np.float32); o[:, 2] = rng.uniform(0.01, 1.99, n)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
prim, tt, b1, b2 = ptamd.trace_rays(o, d)
OL = orc.lib()
import ctypes as C
fp = C.POINTER(C.c_float)
m = 5000
bad = 0
for i in range(m):
    h = OL.orc_trace_closest(o[i].ctypes.data_as(fp), d[i].ctypes.data_as(fp), 0.0, 1e16)
    if h.prim != prim[i] or (h.prim >= 0 and abs(h.t - tt[i]) > 1e-4): bad += 1
print("trace mismatches", bad, "of", m)
for pipe in (0, 1):
    L.pt_set_pipeline(pipe)
    for (w, h, spp) in [(256, 256, 16), (1920, 1080, 16), (1920, 1080, 64)]:
        L.pt_reset_ray_counts()
        L.pt_set_frame(0)
        t = time.time()
        img = ptamd.render_rgb(w, h, spp)
        dt = time.time() - t
        rr, rs = ptamd.ray_counts()
        print(f"pipe {pipe} {w}x{h} spp {spp}: {dt:.3f}s wall, device {L.pt_last_render_ms():.1f} ms, rays {rr}+{rs}, "
              f"{(rr+rs)/(L.pt_last_render_ms()*1e-3)/1e6:.1f} Mrays/s, mean {img.mean(axis=(0,1))}", flush=True)
