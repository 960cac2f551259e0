import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.environ.get("PT_LIB_PATH", os.path.join(ROOT, "rust-pathtracer_amd", "libpt_hip.so")))
fp = C.POINTER(C.c_float)
L.optix_render_rgb.argtypes = [C.c_int] * 4 + [C.c_float, C.c_int, fp]
def render(w, h, spp):
    out = np.empty((h, w, 3), np.float32)
    assert L.optix_render_rgb(w, h, spp, 32, 0.0, 32, out.ctypes.data_as(fp)) == 0
    return out
for mode in (0, 1):
    L.pt_set_mode(mode)
    for spp in (1, 4, 32):
        imgs = []
        for pipeline in (0, 1):
            L.pt_set_pipeline(pipeline); L.pt_set_frame(4)
            imgs.append(render(96, 54, spp))
        d = np.abs(imgs[0] - imgs[1])
        exact = (d == 0).all(axis=2).mean()
        close = (d <= 1e-4 + 1e-4 * np.abs(imgs[1])).all(axis=2).mean()
        print("mode", mode, "spp", spp, "exact", exact, "close", close, "means", imgs[0].mean(), imgs[1].mean())
L.pt_set_mode(1)
imgs = []
for pipeline in (0, 1):
    L.pt_set_pipeline(pipeline); L.pt_set_frame(4)
    imgs.append(render(96, 54, 1))
d = np.abs(imgs[0] - imgs[1])
bad = ~(d <= 1e-4 + 1e-4 * np.abs(imgs[1])).all(axis=2)
ys, xs = np.nonzero(bad)
print("bad pixels", list(zip(xs.tolist(), ys.tolist())))
for x, y in zip(xs[:10], ys[:10]):
    print(x, y, imgs[0][y, x], imgs[1][y, x])
