"""Oracle-only measurement (CPU): what the cadence of the adaptive noise test costs in samples.  The reference tests
after every sample inside launches of 4 (optix_device.cu:1012-1049); the product tests at doubling round boundaries."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc

orc.build()
w, h, spp, mn = 96, 54, 512, 32
print(f"# oracle, {w}x{h}, max {spp} spp, min_adaptive {mn}, converge mode, frame 1")
for thr in (0.01, 0.5, 1.5):
    res = {}
    for name, kw in (("literal in-launch test, launches of 4 (1 thread)", dict(adaptive_deterministic=0, adaptive_round=0, nthreads=1)),
                     ("two-pass test every 4 samples", dict(adaptive_deterministic=1, adaptive_round=4)),
                     ("two-pass test every 32 samples", dict(adaptive_deterministic=1, adaptive_round=32)),
                     ("two-pass test at 32, 64, 128, ... (product default)", dict(adaptive_deterministic=1, adaptive_round=-32))):
        img, _, cnt, rays = orc.render(w, h, spp, frame=1, noise_threshold=thr, min_adaptive=mn, mode=orc.CONVERGE, **kw)
        res[name] = (cnt.mean(), img.mean(), sum(rays), (cnt == mn).mean(), (cnt == spp).mean())
    base = next(iter(res.values()))
    for k, v in res.items():
        print(f"thr {thr:<5} {k:52s} mean spp {v[0]:7.2f} ({v[0] / base[0]:.4f}x)  image mean {v[1]:.5f} ({v[1] / base[1]:.4f}x)  "
              f"rays {v[2]}  stopped at min {v[3]:.3f}  ran to max {v[4]:.3f}")
