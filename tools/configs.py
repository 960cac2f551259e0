"""One-GPU timings of the other benchmark configurations (scratch tool): guided render, 4K Bayer + demosaic."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
import numpy as np, ptamd as gpu
L = gpu.lib()


def synthetic_snapshot(seed=1234, res=(8, 8, 8), lobes_per_region=4):
    rng = np.random.default_rng(seed)
    regions, lobes = [], []
    lo, hi = np.array([-1, -1, 0.0]), np.array([1, 1, 2.0])
    step = (hi - lo) / np.array(res)
    for z in range(res[2]):
        for y in range(res[1]):
            for x in range(res[0]):
                bmin = lo + step * np.array([x, y, z])
                regions.append(list(bmin) + list(bmin + step) + [len(lobes), lobes_per_region])
                for _ in range(lobes_per_region):
                    mu = rng.normal(size=3); mu /= np.linalg.norm(mu)
                    lobes.append(list(mu) + [rng.uniform(1, 50), rng.uniform(0.01, 1), 1, 1, 1])
    return np.array(regions, np.float32), np.array(lobes, np.float32)


def timed(label, fn):
    L.pt_reset_ray_counts()
    t = time.time(); out = fn(); wall = time.time() - t
    rr, rs = gpu.ray_counts()
    ms = L.pt_last_render_ms()
    print(f"{label}: wall {wall:.3f}s device {ms:.1f} ms rays {rr + rs} -> {(rr + rs) / ms / 1e3:.1f} Mrays/s", flush=True)
    return out

gpu.render_rgb(64, 36, 1)
# config 2 for reference
L.pt_set_frame(0); timed("config2 1080p 1024spp", lambda: gpu.render_rgb(1920, 1080, 1024))
# config 3: guiding on, synthetic 8x8x8 field
regions, lobes = synthetic_snapshot()
L.guiding_set_grid_res(8, 8, 8)
gpu.upload_snapshot(regions, lobes)
L.guiding_set_enabled(1)
for pipe in (1, 0):
    L.pt_set_pipeline(pipe); L.pt_set_frame(0)
    timed(f"config3 guided 1080p 256spp pipe {pipe}", lambda: gpu.render_rgb(1920, 1080, 256))
L.guiding_set_enabled(0)
L.pt_set_pipeline(1)
# config 4, one GPU's share: 4K Bayer 512 spp + demosaic
L.pt_set_frame(0)
raw = timed("config4 4K bayer 512spp", lambda: gpu.render_bayer(3840, 2160, 512, pattern=0))
for algo, name in ((0, "gradient"), (1, "bilinear")):
    L.pt_reset_kernel_stats(); L.pt_set_kernel_timing(1)
    t = time.time(); rgb = gpu.demosaic(raw, 0, algo); wall = time.time() - t
    st = gpu.kernel_stats(5)
    print(f"demosaic {name} 4K: wall (with PCIe) {wall * 1e3:.1f} ms device {st['total_ms']:.3f} ms over {st['timed']} timed launches", flush=True)
    L.pt_set_kernel_timing(0)
