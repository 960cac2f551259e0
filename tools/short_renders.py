"""Where the time of short renders goes (scratch tool): 1080p renders of the built-in scene at several sample counts, the whole render
against its one-launch kernel alone, by chunk count: short_renders.py [spp ...]; PT_CHUNKS=n forces the number of chunks."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-pathtracer_amd"))
import ptamd as gpu
L = gpu.lib()
W, H = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
gpu.render_rgb(64, 36, 1)
for spp in [int(a) for a in sys.argv[1:]] or [1024, 256, 128, 64]:
    for rep in range(2):
        L.pt_set_frame(0); L.pt_reset_ray_counts(); L.pt_reset_kernel_stats(); L.pt_set_kernel_timing(1)
        gpu.render_rgb(W, H, spp)
        L.pt_set_kernel_timing(0)
        rr, rs = gpu.ray_counts()
        ms = L.pt_last_render_ms()
        ks = {k: gpu.kernel_stats(i) for i, k in enumerate(gpu.KERNEL_NAMES)}
        kern = " ".join(f"{k} {v['launches']}x{v['total_ms'] / max(v['timed'], 1):.2f}ms" for k, v in ks.items() if v["launches"])
    print(f"{W}x{H} spp {spp}: chunks {L.pt_last_render_chunks()} render {ms:.2f} ms {(rr + rs) / ms / 1e3:.0f} Mrays/s; {kern}", flush=True)
