"""Large synthetic scene, staged pipeline, at several sample counts (scratch tool): does the rate depend on the number of slots in flight?
usage: bigscene_spp.py spheres subdivisions width height spp [spp ...]"""
import sys, time
sys.path.insert(0, "rust-pathtracer_amd"); sys.path.insert(0, "tests")
import ptamd
L = ptamd.lib()
n, sub, w, h = (int(a) for a in sys.argv[1:5])
import os
if os.environ.get('INSTANCED'): L.pt_scene_synthetic_instanced(42, n, sub)  # room + one unit icosphere placed n times
else: L.pt_scene_synthetic(42, n, sub)
ptamd.render_rgb(64, 36, 1)
for spp in (int(a) for a in sys.argv[5:]):
    for pipe in ([int(x) for x in os.environ['PIPES'].split(',')] if os.environ.get('PIPES') else (0, -1)):
        L.pt_set_pipeline(pipe)
        L.pt_reset_ray_counts(); L.pt_set_frame(0); L.pt_reset_kernel_stats(); L.pt_set_kernel_timing(1)
        img = ptamd.render_rgb(w, h, spp)
        L.pt_set_kernel_timing(0)
        rr, rs = ptamd.ray_counts()
        ks = {k: ptamd.kernel_stats(i) for i, k in enumerate(ptamd.KERNEL_NAMES)}
        ext = ks.get("k_extend") or {}
        print(f"pipe {pipe} {w}x{h} spp {spp}: device {L.pt_last_render_ms():.1f} ms rays {rr}+{rs} -> {(rr+rs)/(L.pt_last_render_ms()*1e-3)/1e6:.1f} Mrays/s; "
              + " ".join(f"{k} {v['launches']}x{v['total_ms']/max(v['timed'],1)*1e3:.0f}us" for k, v in ks.items() if v['launches']), flush=True)
