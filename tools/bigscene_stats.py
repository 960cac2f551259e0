"""Per-kernel time of the staged pipeline on the large synthetic scene (scratch tool)."""
import sys, time
sys.path.insert(0, "rust-pathtracer_amd")
import numpy as np, ptamd
L = ptamd.lib()
n, sub, w, h, spp = [int(x) for x in sys.argv[1:6]]
L.pt_scene_synthetic(42, n, sub)
ptamd.render_rgb(64, 36, 1)
print(ptamd.scene_info(), flush=True)
L.pt_set_pipeline(0); L.pt_set_kernel_timing(1); L.pt_reset_kernel_stats(); L.pt_reset_ray_counts(); L.pt_set_frame(0)
ptamd.render_rgb(w, h, spp)
rr, rs = ptamd.ray_counts()
print("device ms", L.pt_last_render_ms(), "rays", rr, rs, "Mrays/s", (rr + rs) / L.pt_last_render_ms() / 1e3)
for k, name in enumerate(ptamd.KERNEL_NAMES):
    st = ptamd.kernel_stats(k)
    if st["launches"]:
        print(f"{name:10s} launches {st['launches']:6d} total {st['total_ms']:9.2f} ms avg {st['total_ms'] / max(st['timed'], 1) * 1e3:9.1f} us")
