"""Demosaic kernel timing at 4K, device-resident, repeated (scratch tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
import numpy as np, torch, ptamd as gpu
L = gpu.lib()
w, h = 3840, 2160
dev = torch.device("cuda:0")
raw = torch.rand(w * h, device=dev, dtype=torch.float32)
out = torch.zeros(w * h * 3, device=dev, dtype=torch.float32)
torch.cuda.synchronize()
for algo, name in ((0, "gradient"), (1, "bilinear")):
    for rep in range(3):
        L.pt_reset_kernel_stats(); L.pt_set_kernel_timing(1)
        for _ in range(10):
            assert L.pt_demosaic_device(raw.data_ptr(), w, h, 0, algo, out.data_ptr()) == 0
        st = gpu.kernel_stats(5)
        L.pt_set_kernel_timing(0)
        ms = st["total_ms"] / max(st["timed"], 1)
        print(f"{name} rep {rep}: {ms * 1e3:.1f} us per demosaic ({st['timed']} timed), {w * h * 16 / ms / 1e6:.1f} GB/s of raw-in + rgb-out", flush=True)
