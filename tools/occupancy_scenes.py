import ctypes as C, os, sys
sys.path.insert(0, "rust-pathtracer_amd")
import ptamd as gpu
L = gpu.lib()
for n, s in ((1, 1), (2, 2), (1, 0), (1, 2)):
    L.pt_scene_synthetic(42, n, s)
    gpu.render_rgb(64, 36, 1)
    info = gpu.scene_info()
    lds = C.c_uint64(0)
    blocks = L.pt_dbg_fused_occupancy(0, C.byref(lds))
    print(n, s, info, "blocks/CU", blocks, "LDS bytes", lds.value)
