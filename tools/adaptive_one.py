"""One adaptive 1080p render for profiling (scratch tool): adaptive_one.py DEFER THRESHOLD [SPP]."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-pathtracer_amd"))
import ptamd as gpu
L = gpu.lib()
defer, thr = int(sys.argv[1]), float(sys.argv[2])
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
L.pt_set_defer_shadow(defer)
gpu.render_rgb(64, 36, 1)
L.pt_set_frame(0); L.pt_reset_ray_counts()
img = gpu.render_rgb(1920, 1080, spp, noise_threshold=thr, min_adaptive_samples=32)
rr, rs = gpu.ray_counts()
ms = L.pt_last_render_ms()
print(f"defer {defer} thr {thr}: {ms:.1f} ms Mrays/s {(rr + rs) / ms / 1e3:.0f} rays {rr}+{rs}", flush=True)
