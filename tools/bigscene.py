"""Large synthetic scene probe (scratch tool): build time, Mrays/s, auto / fused / staged pipelines."""
import sys, time
sys.path.insert(0, "rust-pathtracer_amd"); sys.path.insert(0, "tests")
import numpy as np, ptamd
L = ptamd.lib()
n, sub = int(sys.argv[1]), int(sys.argv[2])
w, h, spp = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
import os
if os.environ.get('INSTANCED'): L.pt_scene_synthetic_instanced(42, n, sub)  # room + one unit icosphere placed n times
else: L.pt_scene_synthetic(42, n, sub)
t = time.time(); img = ptamd.render_rgb(64, 36, 1); print("scene+bvh+first render", round(time.time() - t, 2), "s", ptamd.scene_info(), flush=True)
for pipe in (-1, 1, 0):
    L.pt_set_pipeline(pipe)
    L.pt_reset_ray_counts(); L.pt_set_frame(0)
    t = time.time(); img = ptamd.render_rgb(w, h, spp); dt = time.time() - t
    rr, rs = ptamd.ray_counts()
    print(f"pipe {pipe} {w}x{h} spp {spp}: wall {dt:.2f}s device {L.pt_last_render_ms():.1f} ms rays {rr}+{rs} -> {(rr+rs)/(L.pt_last_render_ms()*1e-3)/1e6:.1f} Mrays/s mean {img.mean():.6f}", flush=True)
