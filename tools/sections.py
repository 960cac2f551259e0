"""Section counters of the fused kernel (profiling build): waves and active lanes per section."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PT_LIB_PATH"] = os.path.join(ROOT, "rust-pathtracer_amd", "libpt_hip_prof.so")
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
import ctypes as C
import numpy as np
import ptamd as gpu
NAMES = ["bounce", "regen", "node", "tri", "shade", "glass", "opaque", "nee", "spec", "diffuse", "sh_node", "sh_tri",
         "shadow", "pop", "sh_pop", "emis_mis", "rr", "fallback"]
w, h, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = gpu.lib()
L.pt_set_pipeline(1)
L.pt_set_frame(0)
L.pt_reset_ray_counts()
if os.environ.get("GUIDED"):  # configuration 3: seeded synthetic field, 8 x 8 x 8 regions of 4 lobes
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
    L.guiding_set_enabled(1)
if os.environ.get("SPHERES"):  # SPHERES="n subdivisions": that many spheres in the room instead of the built-in scene
    n_s, sub = (int(a) for a in os.environ["SPHERES"].split())
    L.pt_scene_synthetic(42, n_s, sub)
gpu.render_rgb(w, h, spp)
buf = (C.c_uint64 * 64)()
n = L.pt_dbg_profile_counters(buf)
rr, rs = gpu.ray_counts()
print("sections", n, "rays", rr, rs, "ms", L.pt_last_render_ms())
bounce_waves = buf[0]
TNAMES = ["loop", "regen", "trace", "shade", "enqueue", "shadow", "epilogue", "node", "tri"]
for k in range(18):
    waves, lanes = buf[2 * k], buf[2 * k + 1]
    print(f"{NAMES[k]:10s} waves {waves:14d}  per-bounce-wave {waves / max(bounce_waves, 1):7.3f}  lanes/wave {lanes / max(waves, 1):6.2f}")
tot = buf[40] + buf[46]
for k, name in enumerate(TNAMES):
    print(f"time {name:10s} {buf[40 + k]:16d} ticks  {buf[40 + k] / max(tot, 1):6.1%} of loop + epilogue")
