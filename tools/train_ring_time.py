"""Cost of capturing training samples during a render (scratch tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
import numpy as np, ptamd as gpu
L = gpu.lib()
for cap in (0, 1 << 18):
    L.guiding_set_train_buffer_size(cap)
    for pipe in (1, 0):
        L.pt_set_pipeline(pipe)
        gpu.render_rgb(1920 + (8 if cap else 0), 1080, 1)  # new size -> new renderer -> ring (re)allocated
        L.pt_set_frame(0); L.pt_reset_ray_counts()
        gpu.render_rgb(1920 + (8 if cap else 0), 1080, 64)
        rr, rs = gpu.ray_counts(); ms = L.pt_last_render_ms()
        print(f"ring {cap} pipeline {pipe}: {ms:.1f} ms, {(rr + rs) / ms / 1e3:.0f} Mrays/s", flush=True)
