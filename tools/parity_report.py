"""Parity numbers of the GPU renderer against the committed oracle goldens (scratch tool)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
import numpy as np, ptamd as gpu
GOLD = os.path.join(ROOT, "tests", "golden")
L = gpu.lib()
def rel_rmse(img, ref): return float(np.sqrt(np.mean(((img - ref) / (ref + 0.01)) ** 2)))
out = {}
g = np.load(os.path.join(GOLD, "strict_32x32_spp4_frame1.npz"))
for pipe, name in ((0, "staged"), (1, "fused")):
    L.pt_set_pipeline(pipe); L.pt_set_mode(gpu.MODE_STRICT); L.pt_set_frame(0); L.pt_reset_ray_counts()
    img = gpu.render_rgb(32, 32, 4)
    rr, rs = gpu.ray_counts()
    out[f"strict_32x32_4spp_{name}"] = {"pixels_within_2e-3": float((np.abs(img - g["rgb"]) <= 2e-3 + 2e-3 * np.abs(g["rgb"])).all(axis=2).mean()),
                                        "rays_gpu": [rr, rs], "rays_oracle": [int(x) for x in g["rays"]]}
L.pt_set_mode(gpu.MODE_CONVERGE); L.pt_set_pipeline(1)
ref = np.load(os.path.join(GOLD, "converge_96x54_spp4096_frame1.npz"))["rgb"]
for spp in (16, 64, 256, 1024, 4096):
    L.pt_set_frame(10)
    img = gpu.render_rgb(96, 54, spp)
    out[f"converge_96x54_{spp}spp"] = {"relRMSE_vs_4096spp_oracle": rel_rmse(img, ref),
                                       "mean_ratio": [float(img[..., c].mean() / ref[..., c].mean()) for c in range(3)]}
gb = np.load(os.path.join(GOLD, "converge_64x64_spp1024_frame2.npz"))
L.pt_set_frame(1)
raw = gpu.render_bayer(64, 64, 1024, pattern=0)
out["bayer_64x64_1024spp"] = {"relRMSE": rel_rmse(raw, gb["bayer"]), "mean_ratio": float(raw.mean() / gb["bayer"].mean())}
print(json.dumps(out, indent=1))
