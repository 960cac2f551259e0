"""Spread of the image mean of guided renders between frames (scratch tool), GPU and oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc, ptamd as gpu
from test_gpu_parity import synthetic_snapshot
L = gpu.lib()
regions, lobes = synthetic_snapshot(seed=1234)
w, h, spp = 64, 48, 512
G, keep = orc.make_guide(regions, lobes, (4, 4, 4))
print("oracle", [float(orc.render(w, h, spp, frame=f, guide=G)[0].mean()) for f in (1, 2, 3)])
L.guiding_set_grid_res(4, 4, 4); gpu.upload_snapshot(regions, lobes); L.guiding_set_enabled(1)
for pipe in (0, 1):
    L.pt_set_pipeline(pipe)
    m = []
    for f in (30, 31, 32, 33, 34, 35):
        L.pt_set_frame(f); m.append(float(gpu.render_rgb(w, h, spp).mean()))
    print("gpu pipe", pipe, m)
