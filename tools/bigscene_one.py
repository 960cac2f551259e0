"""One staged render of the large synthetic scene (scratch tool for parameter sweeps)."""
import sys, time
sys.path.insert(0, "rust-pathtracer_amd")
import ptamd
L = ptamd.lib()
n, sub, w, h, spp = [int(x) for x in sys.argv[1:6]]
L.pt_scene_synthetic(42, n, sub)
ptamd.render_rgb(64, 36, 1)
L.pt_set_pipeline(0); L.pt_reset_ray_counts(); L.pt_set_frame(0)
ptamd.render_rgb(w, h, spp)
rr, rs = ptamd.ray_counts()
print("device ms", round(L.pt_last_render_ms(), 1), "Mrays/s", round((rr + rs) / L.pt_last_render_ms() / 1e3, 1), flush=True)
