"""The same adaptive render several times in one process, optionally after a fixed-count one (scratch tool)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-pathtracer_amd"))
import ptamd as gpu
L = gpu.lib()
gpu.render_rgb(64, 36, 1)
seq = [float(x) for x in sys.argv[1:]] or [0.01, 0.01, 0.01]
for thr in seq:
    L.pt_set_frame(0); L.pt_reset_ray_counts()
    gpu.render_rgb(1920, 1080, 1024, noise_threshold=thr, min_adaptive_samples=32)
    rr, rs = gpu.ray_counts()
    ms = L.pt_last_render_ms()
    print(f"thr {thr}: {ms:.1f} ms Mrays/s {(rr + rs) / ms / 1e3:.0f} chunks {L.pt_last_render_chunks()}", flush=True)
