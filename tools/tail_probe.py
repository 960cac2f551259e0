"""How a one-launch render ends (profiling build): when the first wave starts, when the first and the last wave find the slot queue
empty, when the last wave ends: tail_probe.py [spp ...]; W, H set the image size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PT_LIB_PATH"] = os.path.join(ROOT, "rust-pathtracer_amd", "libpt_hip_prof.so")
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
import ctypes as C
import ptamd as gpu
L = gpu.lib()
W, H = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
L.pt_set_pipeline(1)
gpu.render_rgb(64, 36, 1)
buf = (C.c_uint64 * 64)()
L.pt_dbg_profile_counters(buf)
for spp in [int(a) for a in sys.argv[1:]] or [1024, 128]:
    L.pt_set_frame(0)
    gpu.render_rgb(W, H, spp)
    L.pt_dbg_profile_counters(buf)
    big = 1 << 62
    t0, e0, e1, t1 = big - buf[56], big - buf[57], buf[58], buf[59]
    ms = lambda a, b: (b - a) / 1e5
    print(f"{W}x{H} spp {spp} chunks {L.pt_last_render_chunks()}: render {L.pt_last_render_ms():.2f} ms; kernel {ms(t0, t1):.2f} ms: first wave finds the queue empty "
          f"after {ms(t0, e0):.2f} ms, the last after {ms(t0, e1):.2f} ms, last wave ends {ms(e0, t1):.2f} ms after the first found it empty", flush=True)
