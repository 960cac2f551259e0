"""Section counters of the staged traversal kernels on a synthetic scene (profiling build):
loop iterations, node visits and item tests per ray, and the lanes that take part in each.
usage: sections_staged.py spheres subdivisions width height spp"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PT_LIB_PATH"] = os.path.join(ROOT, "rust-pathtracer_amd", "libpt_hip_prof.so")
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
import ctypes as C
import ptamd as gpu
n, sub, w, h, spp = (int(a) for a in sys.argv[1:6])
L = gpu.lib()
L.pt_scene_synthetic(42, n, sub)
L.pt_set_pipeline(0)
gpu.render_rgb(64, 36, 1)
buf = (C.c_uint64 * 64)()
L.pt_dbg_profile_counters(buf)
L.pt_set_frame(0)
L.pt_reset_ray_counts()
gpu.render_rgb(w, h, spp)
L.pt_dbg_profile_counters(buf)
rr, rs = gpu.ray_counts()
print(gpu.scene_info(), "rays", rr, rs, "ms", L.pt_last_render_ms())
for name, rays, it, nd, tr in (("extend", rr, 0, 2, 3), ("occlude", rs, 12, 10, 11)):
    iw, il = buf[2 * it], buf[2 * it + 1]
    nw, nl = buf[2 * nd], buf[2 * nd + 1]
    tw, tl = buf[2 * tr], buf[2 * tr + 1]
    print(f"{name}: {iw * 64 / rays:.2f} wave-iteration lane-slots per ray at {il / max(iw, 1):.1f} busy lanes; "
          f"node visits {nl / rays:.2f} per ray in {nw / max(iw, 1):.2f} of the iterations at {nl / max(nw, 1):.1f} lanes; "
          f"item tests {tl / rays:.2f} per ray in {tw / max(iw, 1):.2f} of the iterations at {tl / max(tw, 1):.1f} lanes")
