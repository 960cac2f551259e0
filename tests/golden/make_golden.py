"""Generates the committed golden vectors from the CPU oracle (oracle/liborc.so).

The reference ships no vectors beyond its four cfa_at tests (src/main.rs:566-601) and
cannot be built or run here (nvcc / OptiX / OpenPGL / cargo are absent), so these
fixtures pin the oracle against itself over time and give the GPU tests fixed targets.
Run:  python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402

L = orc.lib()
fp = C.POINTER(C.c_float)


def f32(a):
    return np.asarray(a, np.float32)


def main():
    out = {}
    # RNG: first 16 outputs for fixed seeds (a3)
    out["pcg"] = {}
    for seed in (0, 1, 0x853c49e6748fea9b, L.orc_pixel_seed(1, 0), L.orc_pixel_seed(2, 123456)):
        s = C.c_uint64(seed)
        u = [int(L.orc_pcg(C.byref(s))) for _ in range(16)]
        s = C.c_uint64(seed)
        f = [float(L.orc_rnd(C.byref(s))) for _ in range(16)]
        out["pcg"][str(seed)] = {"u32": u, "f32": f}
    out["pixel_seed"] = [[f, d, int(L.orc_pixel_seed(f, d))] for f in (0, 1, 2, 77) for d in (0, 1, 1919, 2073599)]
    # blue noise / jitter (a4)
    bn = []
    for x, y, s, fr in [(0, 0, 0, 1), (1, 0, 0, 1), (0, 1, 3, 1), (17, 5, 2, 2), (1919, 1079, 1023, 1), (640, 360, 4095, 7)]:
        uv = (C.c_float * 2)()
        jt = (C.c_float * 2)()
        L.orc_blue_noise(x, y, s, fr, uv)
        L.orc_olpf_jitter(x, y, s, fr, jt)
        bn.append({"xysf": [x, y, s, fr], "uv": list(uv), "jitter": list(jt)})
    out["blue_noise"] = bn
    # camera (a5)
    cams = []
    for (w, h) in [(256, 256), (1920, 1080), (3840, 2160)]:
        eye, u, v, ww = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
        L.orc_camera_default(w, h, eye, u, v, ww)
        dirs = []
        for (x, y) in [(0, 0), (w - 1, 0), (0, h - 1), (w - 1, h - 1), (w // 2, h // 2)]:
            for jit in [(0.0, 0.0), (0.75, -0.5)]:
                d = (C.c_float * 3)()
                L.orc_camera_dir(x, y, (C.c_float * 2)(*jit), w, h, u, v, ww, d)
                dirs.append({"xy": [x, y], "jitter": list(jit), "dir": list(d)})
        cams.append({"w": w, "h": h, "eye": list(eye), "u": list(u), "v": list(v), "w_axis": list(ww), "dirs": dirs})
    out["camera"] = cams
    # sample_ggx_vndf (a8)
    rng = np.random.default_rng(11)
    ggx = []
    for _ in range(8):
        N = rng.normal(size=3); N /= np.linalg.norm(N)
        V = rng.normal(size=3); V /= np.linalg.norm(V)
        if np.dot(N, V) < 0: V = -V
        u2 = rng.uniform(size=2)
        alpha = float(rng.uniform(0.01, 0.5))
        Lv, H, pdf = (C.c_float * 3)(), (C.c_float * 3)(), C.c_float()
        L.orc_sample_ggx_vndf((C.c_float * 2)(*u2), alpha, orc.f3(N), orc.f3(V), Lv, H, C.byref(pdf))
        ggx.append({"u": f32(u2).tolist(), "alpha": float(np.float32(alpha)), "N": f32(N).tolist(), "V": f32(V).tolist(),
                    "L": list(Lv), "H": list(H), "pdf": pdf.value})
    out["ggx_vndf"] = ggx
    # vMF (a13)
    vm = []
    for kappa in (0.0, 1e-5, 0.5, 5.0, 50.0):
        mu = rng.normal(size=3); mu /= np.linalg.norm(mu)
        u2 = rng.uniform(size=2)
        o = (C.c_float * 3)()
        L.orc_vmf_sample(orc.f3(mu), kappa, (C.c_float * 2)(*u2), o)
        pdf = L.orc_vmf_pdf(orc.f3(mu), kappa, o)
        vm.append({"mu": f32(mu).tolist(), "kappa": kappa, "u": f32(u2).tolist(), "dir": list(o), "pdf": pdf})
    out["vmf"] = vm
    # ray hits on the built-in scene (a7)
    hits = []
    rays = [((0, 3, 1), (0, -1, 0)), ((0, 3, 1), (0.2, -1, 0.3)), ((0, 3, 1), (-0.25, -1, -0.3)),
            ((0.5, -0.5, 1.9), (0, 0, -1)), ((0, 0, 1), (0, 1, 0)), ((0, 0, 1), (0.1, 0.2, 1)),
            ((0.5, -0.5, 0.3), (1, 0.3, 0.1)), ((-0.9, 0.9, 0.1), (1, -1, 0.5))]
    for o, d in rays:
        d = np.asarray(d, np.float64); d /= np.linalg.norm(d)
        h = L.orc_trace_closest(orc.f3(o), orc.f3(d), 0.0, 1e16)
        hits.append({"o": f32(o).tolist(), "d": f32(d).tolist(), "prim": h.prim, "t": h.t, "b1": h.b1, "b2": h.b2})
    out["hits"] = hits
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(out, f, indent=1)

    # images: strict 32x32 (one 4-sample launch) and a converged 96x54 mean image
    rgb, bayer, _, rays = orc.render(32, 32, 4, frame=1, mode=orc.STRICT)
    np.savez_compressed(os.path.join(HERE, "strict_32x32_spp4_frame1.npz"), rgb=rgb, bayer=bayer, rays=np.array(rays))
    rgb, bayer, _, rays = orc.render(96, 54, 4096, frame=1, mode=orc.CONVERGE)
    np.savez_compressed(os.path.join(HERE, "converge_96x54_spp4096_frame1.npz"), rgb=rgb.astype(np.float32), rays=np.array(rays))
    rgb, bayer, _, rays = orc.render(64, 64, 1024, frame=2, mode=orc.CONVERGE, pattern=0)
    np.savez_compressed(os.path.join(HERE, "converge_64x64_spp1024_frame2.npz"), rgb=rgb, bayer=bayer, rays=np.array(rays))
    print("golden written")


if __name__ == "__main__":
    main()
