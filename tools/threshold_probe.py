"""What the thresholded quantities of the GPU parity tests measure on this build (scratch tool): the test
thresholds are set to about 1.3 times these figures."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
import ptamd as gpu

GOLD = os.path.join(ROOT, "tests", "golden")
L = gpu.lib()
out = {}


def rel_rmse(img, ref):
    return float(np.sqrt(np.mean(((img - ref) / (ref + 0.01)) ** 2)))


# adaptive sampling against the oracle (test_adaptive_sampling_matches_oracle)
w, h, spp, thr, mn, rnd = 96, 32, 256, 1.5, 32, 32
for pipeline in (0, 1):
    for defer in (0, 1):
        L.pt_set_pipeline(pipeline)
        L.pt_set_defer_shadow(defer)
        L.pt_set_adaptive_chunks(1)
        for cadence in (rnd, -32):
            L.pt_set_adaptive_round(cadence)
            L.pt_set_frame(0)
            img = gpu.render_rgb(w, h, spp, noise_threshold=thr, min_adaptive_samples=mn)
            counts, rounds = gpu.sample_counts(w, h)
            o_img, _, o_counts, _ = orc.render(w, h, spp, frame=1, noise_threshold=thr, min_adaptive=mn, mode=orc.CONVERGE,
                                               adaptive_deterministic=1, adaptive_round=cadence)
            out[f"adaptive_pipe{pipeline}_defer{defer}_round{cadence}"] = {
                "same_counts": float((counts == o_counts).mean()),
                "mean_spp": [float(counts.mean()), float(o_counts.mean())],
                "close_2e-2": float((np.abs(img - o_img) <= 2e-3 + 2e-2 * np.abs(o_img)).all(axis=2).mean()),
                "mean_ratio": float(img.mean() / o_img.mean())}
L.pt_set_adaptive_round(-32)
L.pt_set_adaptive_chunks(0)
L.pt_set_defer_shadow(0)
L.pt_set_pipeline(1)

# smoke()'s quantities
for pipe in (0, 1):
    L.pt_set_pipeline(pipe)
    L.pt_set_mode(gpu.MODE_CONVERGE)
    L.pt_set_frame(0)
    img = gpu.render_rgb(64, 48, 64)
    ref = orc.render(64, 48, 64, frame=1, mode=orc.CONVERGE)[0]
    L.pt_set_mode(gpu.MODE_STRICT)
    L.pt_set_frame(0)
    img4 = gpu.render_rgb(64, 48, 4)
    ref4 = orc.render(64, 48, 4, frame=1, mode=orc.STRICT)[0]
    out[f"smoke_pipe{pipe}"] = {"mean_rel": [float(x) for x in np.abs(img.mean(axis=(0, 1)) - ref.mean(axis=(0, 1))) / ref.mean(axis=(0, 1))],
                                "strict_close": float((np.abs(img4 - ref4) <= 2e-3 + 2e-3 * np.abs(ref4)).all(axis=2).mean())}
L.pt_set_mode(gpu.MODE_CONVERGE)
L.pt_set_pipeline(1)

# the rays that start on a plane (test_rays_that_start_on_a_plane_keep_their_small_known_bias)
ref = np.load(os.path.join(GOLD, "converge_96x54_spp4096_frame1.npz"))["rgb"].astype(np.float64)
acc = np.zeros_like(ref)
per = []
for f in (10, 20, 30, 40):
    L.pt_set_frame(f)
    img = gpu.render_rgb(96, 54, 1024)
    per.append([float(x) for x in img.mean(axis=(0, 1)) / ref.mean(axis=(0, 1))])
    acc += img
out["on_plane_mean_ratio"] = {"per_frame": per, "four_frames": [float(x) for x in (acc / 4).mean(axis=(0, 1)) / ref.mean(axis=(0, 1))]}

# 1080p blocks against the golden (test_full_size_1080p_properties)
L.pt_set_frame(0)
img = gpu.render_rgb(1920, 1080, 32)
from test_gpu_parity import tent_downsample
ref32 = ref.astype(np.float32)
out["blocks_1080p_32spp_relrmse"] = {"box": rel_rmse(img.reshape(54, 20, 96, 20, 3).mean(axis=(1, 3)), ref32),
                                     "tent_inner": rel_rmse(tent_downsample(img, 20)[1:-1, 1:-1], ref32[1:-1, 1:-1])}

# the Bayer configuration as a chain at its own size (test_bayer_render_and_demosaic_chain_at_full_size)
w, h, spp = 3840, 2160, 256
L.pt_set_frame(1)
raw = gpu.render_bayer(w, h, spp, pattern=0)
L.pt_set_frame(1)
rgb = gpu.render_rgb(w, h, spp)
dem = gpu.demosaic(raw, 0, 0)
refi = ref32[1:-1, 1:-1]
lit = refi.sum(axis=2) > 0.05
for name, img in (("rgb", rgb), ("demosaic", dem)):
    b = tent_downsample(img, 40)[1:-1, 1:-1]
    rel = np.abs(b - refi)[lit] / (refi[lit] + 0.01)
    out[f"config4_{name}_tent"] = {"median_rel": float(np.median(rel)), "p99_rel": float(np.percentile(rel, 99)),
                                   "relrmse": rel_rmse(b, refi), "mean_ratio": [float(x) for x in img.mean(axis=(0, 1)) / ref.mean(axis=(0, 1))]}
print(json.dumps(out, indent=1))
