"""One guided render (configuration 3: the seeded 8 x 8 x 8 field of tools/configs.py, 1080p): guided_render.py [spp] (scratch tool; the
command the counter passes of tools/pmc_guided.sh run)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
import numpy as np, ptamd as gpu
L = gpu.lib()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(1234)
regions, lobes = [], []
step = np.array([2, 2, 2.0]) / 8
for z in range(8):
    for y in range(8):
        for x in range(8):
            bmin = np.array([-1, -1, 0.0]) + step * np.array([x, y, z])
            regions.append(list(bmin) + list(bmin + step) + [len(lobes), 4])
            for _ in range(4):
                mu = rng.normal(size=3); mu /= np.linalg.norm(mu)
                lobes.append(list(mu) + [rng.uniform(1, 50), rng.uniform(0.01, 1), 1, 1, 1])
L.guiding_set_grid_res(8, 8, 8)
gpu.upload_snapshot(np.array(regions, np.float32), np.array(lobes, np.float32))
L.guiding_set_enabled(int(os.environ.get("GUIDED", "1")))
L.pt_set_pipeline(1); L.pt_set_frame(0); L.pt_reset_ray_counts()
gpu.render_rgb(1920, 1080, spp)
rr, rs = gpu.ray_counts()
ms = L.pt_last_render_ms()
print(f"guided {L.guiding_get_enabled() if hasattr(L, 'guiding_get_enabled') else '?'} 1080p {spp} spp: {ms:.1f} ms, {(rr + rs) / ms / 1e3:.1f} Mrays/s")
