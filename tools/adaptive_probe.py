import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-pathtracer_amd"))
import numpy as np, ptamd as gpu, orc
L = gpu.lib()
w, h, spp, thr, mn, rnd = 96, 32, 256, 1.5, 32, 32
for pipeline in (0, 1):
    L.pt_set_pipeline(pipeline); L.pt_set_mode(gpu.MODE_CONVERGE); L.pt_set_adaptive_round(rnd); L.pt_set_frame(0); L.pt_reset_ray_counts()
    img = gpu.render_rgb(w, h, spp, noise_threshold=thr, min_adaptive_samples=mn)
    counts, rounds = gpu.sample_counts(w, h)
    rr, rs = gpu.ray_counts()
    o_img, _, o_counts, o_rays = orc.render(w, h, spp, frame=1, noise_threshold=thr, min_adaptive=mn, mode=orc.CONVERGE, adaptive_deterministic=1, adaptive_round=rnd)
    print("pipeline", pipeline, "rounds", rounds, "same", (counts == o_counts).mean(), "mean counts", counts.mean(), o_counts.mean(),
          "rays", rr + rs, sum(o_rays), "img mean", img.mean(), o_img.mean())
    print(" uniq", np.unique(counts, return_counts=True), np.unique(o_counts, return_counts=True))
    close = (np.abs(img - o_img) <= 2e-3 + 2e-2 * np.abs(o_img)).all(axis=2).mean()
    print(" close", close)
    L.pt_set_frame(0)
    plain = gpu.render_rgb(w, h, spp)
    o_plain = orc.render(w, h, spp, frame=1)[0]
    print(" plain mean", plain.mean(), o_plain.mean(), (np.abs(plain - o_plain) <= 2e-3 + 2e-2 * np.abs(o_plain)).all(axis=2).mean())
# timing at 1080p
import time
for thr in (0.0, 1.0, 2.0):
    L.pt_set_pipeline(1); L.pt_set_frame(0); L.pt_reset_ray_counts()
    t = time.time(); gpu.render_rgb(1920, 1080, 1024, noise_threshold=thr, min_adaptive_samples=32); dt = time.time() - t
    c, r = gpu.sample_counts(1920, 1080)
    rr, rs = gpu.ray_counts()
    print("1080p thr", thr, "time", dt, "ms", L.pt_last_render_ms(), "rounds", r, "mean spp", c.mean(), "Mrays/s", (rr + rs) / L.pt_last_render_ms() / 1e3)
