"""Resident blocks per CU of the fused kernels as the runtime counts them (scratch tool)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-pathtracer_amd"))
import ptamd as gpu
import torch
L = gpu.lib()
gpu.render_rgb(64, 36, 1)
p = torch.cuda.get_device_properties(0)
print("device", p.name, "CUs", p.multi_processor_count, "shared per block", getattr(p, "shared_memory_per_block", None),
      "shared per CU", getattr(p, "shared_memory_per_multiprocessor", None))
for defer in (0, 1):
    L.pt_set_defer_shadow(defer)
    for adaptive in (0, 1):
        lds = C.c_uint64(0)
        n = L.pt_dbg_fused_occupancy(adaptive, C.byref(lds))
        print(f"defer {defer} adaptive {adaptive}: {n} blocks per CU, {lds.value} bytes of LDS per block")
L.pt_set_defer_shadow(-1)
