"""Adaptive render timings at 1080p (scratch tool)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-pathtracer_amd"))
import numpy as np, ptamd as gpu
L = gpu.lib()
gpu.render_rgb(64, 36, 1)
for thr in (0.0, 0.01, 1.0, 2.0):
    L.pt_set_frame(0); L.pt_reset_ray_counts()
    gpu.render_rgb(1920, 1080, 1024, noise_threshold=thr, min_adaptive_samples=32)
    c, r = gpu.sample_counts(1920, 1080)
    rr, rs = gpu.ray_counts()
    ms = L.pt_last_render_ms()
    print(f"thr {thr}: {ms:.1f} ms rounds {r} mean spp {c.mean():.1f} Mrays/s {(rr + rs) / ms / 1e3:.0f}", flush=True)
