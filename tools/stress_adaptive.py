"""Random resolutions / sample counts / round settings through the adaptive and fixed render paths (scratch tool):
every image finite, per-pixel counts within bounds and equal to spp where nothing can stop."""
import sys
sys.path.insert(0, "rust-pathtracer_amd")
import numpy as np, ptamd as gpu
L = gpu.lib()
rng = np.random.default_rng(1)
for it in range(40):
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    spp = int(rng.integers(1, 300))
    rnd = int(rng.choice([-32, -8, 5, 12, 32, 40]))
    chunks = int(rng.choice([0, 0, 1, 2, 16]))
    pipe = int(rng.choice([-1, 0, 1]))
    thr = float(rng.choice([1e-9, 0.5, 1.5, 1e9]))
    mn = int(rng.choice([1, 8, 32, 100]))
    L.pt_set_adaptive_round(rnd); L.pt_set_adaptive_chunks(chunks); L.pt_set_pipeline(pipe)
    img = gpu.render_rgb(w, h, spp, noise_threshold=thr, min_adaptive_samples=mn)
    c, rounds = gpu.sample_counts(w, h)
    assert np.isfinite(img).all() and img.min() >= 0, (w, h, spp)
    assert c.min() >= min(1, spp) and c.max() <= spp, (w, h, spp, rnd, chunks, c.min(), c.max())
    if thr == 1e-9:
        lit = img.sum(axis=2) > 0
        assert (c[lit] == spp).all(), (w, h, spp, rnd, chunks)
    fixed = gpu.render_rgb(w, h, spp)
    c2, _ = gpu.sample_counts(w, h)
    assert (c2 == spp).all() and np.isfinite(fixed).all()
    print(it, w, h, spp, rnd, chunks, pipe, thr, mn, "rounds", rounds, "mean count", round(float(c.mean()), 1), flush=True)
print("ok")
