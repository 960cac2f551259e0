"""8K render smoke check (scratch tool): index arithmetic and buffer sizes at 33 M pixels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
import numpy as np, ptamd as gpu
L = gpu.lib()
for pipe in (1, 0):
    L.pt_set_pipeline(pipe); L.pt_set_frame(0); L.pt_reset_ray_counts()
    img = gpu.render_rgb(7680, 4320, 8)
    rr, rs = gpu.ray_counts()
    print("pipeline", pipe, "finite", bool(np.isfinite(img).all()), "mean", float(img.mean()), "last row mean", float(img[-1].mean()),
          "ms", round(L.pt_last_render_ms(), 1), "Mrays/s", round((rr + rs) / L.pt_last_render_ms() / 1e3), flush=True)
    lo = gpu.render_rgb(960, 540, 8)
    blk = img.reshape(540, 8, 960, 8, 3).mean(axis=(1, 3))
    print("  8x8 block means vs 960x540 render: mean ratio", float(blk.mean() / lo.mean()))
raw = gpu.render_bayer(7680, 4320, 4)
rgb = gpu.demosaic(raw, 0, 0)
print("bayer+demosaic 8K finite", bool(np.isfinite(rgb).all()), float(rgb.mean()))
