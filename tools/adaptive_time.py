"""Adaptive render timings at 1080p (scratch tool): the exact fused kernel (shadow rays traced in place) against the
kernel with the per-wave shadow queue, at the reference's threshold and at thresholds that stop lit pixels."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-pathtracer_amd"))
import numpy as np, ptamd as gpu
L = gpu.lib()
gpu.render_rgb(64, 36, 1)
for defer in (0, 1):
    L.pt_set_defer_shadow(defer)
    for thr in (0.0, 0.01, 1.0, 2.0):
        for rep in range(2):
            L.pt_set_frame(0); L.pt_reset_ray_counts()
            img = gpu.render_rgb(1920, 1080, 1024, noise_threshold=thr, min_adaptive_samples=32)
        c, r = gpu.sample_counts(1920, 1080)
        rr, rs = gpu.ray_counts()
        ms = L.pt_last_render_ms()
        print(f"defer {defer} thr {thr}: {ms:.1f} ms rounds {r} mean spp {c.mean():.1f} Mrays/s {(rr + rs) / ms / 1e3:.0f} image mean {img.mean():.6f}", flush=True)
L.pt_set_defer_shadow(-1)
