"""Adaptive renders of a synthetic scene that runs staged, with the chunk count held to what the small-scene rule
allowed and with the staged rule (scratch tool).  usage: adaptive_big.py spheres subdivisions width height spp threshold"""
import sys
sys.path.insert(0, "rust-pathtracer_amd")
import ptamd
L = ptamd.lib()
n, sub, w, h, spp = (int(a) for a in sys.argv[1:6])
thr = float(sys.argv[6])
L.pt_scene_synthetic(42, n, sub)
ptamd.render_rgb(64, 36, 1)
for cap in (1, 4, 0):
    L.pt_set_adaptive_chunks(cap)
    for rep in range(2):
        L.pt_reset_ray_counts(); L.pt_set_frame(3)
        img = ptamd.render_rgb(w, h, spp, noise_threshold=thr, min_adaptive_samples=32)
        rr, rs = ptamd.ray_counts()
        counts, rounds = ptamd.sample_counts(w, h)
        print(f"cap {cap} chunks {L.pt_last_render_chunks()}: device {L.pt_last_render_ms():.1f} ms, mean spp {counts.mean():.1f}, rounds {rounds}, "
              f"{(rr + rs) / (L.pt_last_render_ms() * 1e-3) / 1e6:.1f} Mrays/s, image mean {img.mean():.5f}", flush=True)
