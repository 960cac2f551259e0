"""ptamd -- Python host binding of libpt_hip.so, the MI355X path-tracing backend.

This is the ctypes counterpart of the reference's Rust FFI block (src/main.rs:9-32,
src/guiding.rs:31-68): the same symbols with the same argument meaning, plus the pt_*
additions declared in include/pt_api.h.  There is no CPU fallback: importing works
anywhere (so that the symbol table can be inspected), but every compute call needs the
HIP library and a GPU, and raises if the library is missing.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("PT_LIB_PATH", os.path.join(ROOT, "libpt_hip.so"))  # override: A/B builds of the same ABI

MODE_STRICT, MODE_CONVERGE = 0, 1
PIPELINE_STAGED, PIPELINE_FUSED = 0, 1
K_GENERATE, K_EXTEND, K_SHADE, K_OCCLUDE, K_BOUNCE, K_DEMOSAIC = range(6)
KERNEL_NAMES = ["generate", "extend", "shade", "occlude", "bounce", "demosaic"]


class PglRegion(C.Structure):  # src/guiding.rs:13-20
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("lobe_ofs", C.c_uint32), ("lobe_num", C.c_uint32)]


class PglLobe(C.Structure):  # src/guiding.rs:21-28
    _fields_ = [("mu", C.c_float * 3), ("kappa", C.c_float), ("weight", C.c_float), ("rgb", C.c_float * 3)]


class TrainSample(C.Structure):  # src/guiding.rs:5-11
    _fields_ = [("position", C.c_float * 3), ("dir_in", C.c_float * 3), ("contrib", C.c_float * 3),
                ("is_delta", C.c_uint32)]


class PathRec(C.Structure):  # pt_path_rec
    _fields_ = [("radiance", C.c_float * 3), ("throughput", C.c_float * 3), ("origin", C.c_float * 3),
                ("direction", C.c_float * 3), ("seed", C.c_uint64), ("depth", C.c_int), ("done", C.c_int),
                ("prev_pdf_bsdf", C.c_float), ("prev_pdf_valid", C.c_int), ("px", C.c_int), ("py", C.c_int),
                ("sample", C.c_int), ("rng_dim", C.c_int)]


_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_u32p = C.POINTER(C.c_uint32)
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)  # pt_reduce_fn
_u64p = C.POINTER(C.c_uint64)

# name -> (restype, argtypes); also the list of symbols include/pt_api.h declares.
SIGNATURES = {
    "optix_render_rgb": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _fp]),
    "optix_render_bayer_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _fp]),
    "optix_alloc_host": (C.c_void_p, [C.c_size_t]),
    "optix_free_host": (None, [C.c_void_p]),
    "optix_synchronize": (None, []),
    "optix_ctx_create": (C.c_void_p, [C.c_int, C.c_int]),
    "optix_ctx_destroy": (None, [C.c_void_p]),
    "optix_ctx_set_bayer": (None, [C.c_void_p, C.c_int]),
    "optix_ctx_set_camera": (None, [C.c_void_p] + [C.c_float] * 10),
    "guiding_upload_snapshot": (None, [C.POINTER(PglRegion), C.c_uint32, C.POINTER(PglLobe), C.c_uint32]),
    "guiding_set_enabled": (None, [C.c_int]),
    "guiding_set_train_buffer_size": (None, [C.c_uint32]),
    "guiding_map_train_samples": (None, [C.POINTER(C.POINTER(TrainSample)), _u32p]),
    "guiding_reset_train_write_idx": (None, []),
    "guiding_set_grid_res": (None, [C.c_int, C.c_int, C.c_int]),
    "pgl_dev_create": (C.c_void_p, [C.c_int]),
    "pgl_dev_destroy": (None, [C.c_void_p]),
    "pgl_field_create": (C.c_void_p, [C.c_void_p, _fp, _fp]),
    "pgl_field_destroy": (None, [C.c_void_p]),
    "pgl_samples_create": (C.c_void_p, []),
    "pgl_samples_destroy": (None, [C.c_void_p]),
    "pgl_samples_add_surface": (None, [C.c_void_p, _fp, _fp, _fp, C.c_int]),
    "pgl_field_update": (None, [C.c_void_p, C.c_void_p]),
    "pt_pgl_samples_add_batch": (None, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "pt_pgl_field_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int),
                                      C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "pt_pgl_field_update_binned": (None, [C.c_void_p, C.c_void_p]),
    "pt_guiding_bin_train_samples_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pt_pgl_field_update_from_ring": (C.c_int, [C.c_void_p]),
    "pgl_field_snapshot": (None, [C.c_void_p, C.POINTER(C.POINTER(PglRegion)), _u32p,
                                  C.POINTER(C.POINTER(PglLobe)), _u32p]),
    "pt_set_mode": (None, [C.c_int]),
    "pt_get_mode": (C.c_int, []),
    "pt_set_pipeline": (None, [C.c_int]),
    "pt_get_pipeline": (C.c_int, []),
    "pt_set_device": (None, [C.c_int]),
    "pt_set_device_count": (None, [C.c_int]),
    "pt_get_device_count": (C.c_int, []),
    "pt_shard_range": (None, [C.c_int, C.c_int, C.c_int, _ip, _ip]),
    "pt_set_frame": (None, [C.c_int]),
    "pt_get_frame": (C.c_int, []),
    "pt_render_accumulate_device": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pt_render_accumulate_device_noalloc": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pt_render_adaptive_accumulate_device": (C.c_int, [C.c_int] * 6 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "pt_render_adaptive_shared_device": (C.c_int, [C.c_int] * 7 + [C.c_float, C.c_int, C.c_void_p, REDUCE_FN, C.c_void_p]),
    "pt_mosaic_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "pt_demosaic_f32": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "pt_demosaic_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pt_write_exr": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _fp, C.c_float]),
    "pt_write_png": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _fp, C.c_float, C.c_int]),
    "pt_read_exr": (C.c_int, [C.c_char_p, _ip, _ip, _fp, C.c_size_t]),
    "pt_get_ray_counts": (None, [_u64p, _u64p]),
    "pt_reset_ray_counts": (None, []),
    "pt_set_kernel_timing": (None, [C.c_int]),
    "pt_get_kernel_stats": (None, [C.c_int, C.POINTER(C.c_double), _u64p, _u64p, _u64p]),
    "pt_reset_kernel_stats": (None, []),
    "pt_last_render_ms": (C.c_double, []),
    "pt_last_render_chunks": (C.c_int, []),
    "pt_dbg_chunk_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulonglong]),
    "pt_dbg_profile_counters": (C.c_int, [C.POINTER(C.c_uint64)]),
    "pt_dbg_fused_occupancy": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "pt_set_defer_shadow": (None, [C.c_int]),
    "pt_set_one_launch": (None, [C.c_int]),
    "pt_get_one_launch": (C.c_int, []),
    "pt_get_defer_shadow": (C.c_int, []),
    "pt_set_adaptive_round": (None, [C.c_int]),
    "pt_get_adaptive_round": (C.c_int, []),
    "pt_set_adaptive_chunks": (None, [C.c_int]),
    "pt_get_adaptive_chunks": (C.c_int, []),
    "pt_get_sample_counts": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pt_scene_cornell": (None, []),
    "pt_scene_synthetic": (None, [C.c_uint32, C.c_int, C.c_int]),
    "pt_scene_triangles": (C.c_int, [_fp, C.c_int, _u32p, C.c_int, _fp]),
    "pt_set_bvh_pairing": (None, [C.c_int]),
    "pt_scene_load_obj": (C.c_int, [C.c_char_p]),
    "pt_scene_counts": (None, [_ip, _ip]),
    "pt_scene_instanced_begin": (C.c_int, []),
    "pt_scene_add_mesh": (C.c_int, [_fp, C.c_int, _u32p, C.c_int, _fp]),
    "pt_scene_add_instance": (C.c_int, [C.c_int, _fp, _fp]),
    "pt_scene_instanced_commit": (C.c_int, []),
    "pt_scene_flatten": (C.c_int, []),
    "pt_scene_instance_count": (C.c_int, []),
    "pt_scene_synthetic_instanced": (None, [C.c_uint32, C.c_int, C.c_int]),
    "pt_trace_rays_instances": (C.c_int, [C.c_int, _fp, _fp, _fp, _fp, C.c_int, _ip, _fp, _fp, _fp, _ip]),
    "pt_scene_save_obj": (C.c_int, [C.c_char_p]),
    "pt_scene_set_light": (None, [_fp, _fp, _fp, _fp]),
    "pt_scene_info": (None, [_ip, _ip, _ip, C.POINTER(C.c_size_t)]),
    "pt_scene_get": (None, [_fp, _u32p, _fp, _fp]),
    "pt_dbg_pcg": (C.c_int, [C.c_uint64, C.c_int, _u32p, _fp]),
    "pt_dbg_blue_noise": (C.c_int, [C.c_int, _ip, _fp, _fp]),
    "pt_dbg_camera_dirs": (C.c_int, [C.c_int, C.c_int, C.c_int, _ip, _fp, _fp]),
    "pt_trace_rays": (C.c_int, [C.c_int, _fp, _fp, _fp, _fp, C.c_int, _ip, _fp, _fp, _fp]),
    "pt_dbg_shade": (C.c_int, [C.c_int, C.POINTER(PathRec), _ip, _fp, _fp, C.c_int, C.c_int, _ip]),
    "pt_dbg_guiding": (C.c_int, [C.c_int, _fp, _fp, _fp, _ip, _ip, _fp, _fp, _fp]),
    "pt_version": (C.c_char_p, []),
}

_lib = None


def lib():
    """Load libpt_hip.so (built in-tree by `make -C rust-pathtracer_amd`).  No fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C {ROOT}` (hipcc, gfx950). "
                "There is no CPU fallback for the render path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if "PT_LIB_PATH" in os.environ and not hasattr(L, name):
                continue  # an older build of the ABI under comparison
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(_fp)


def _i(a):
    return a.ctypes.data_as(_ip)


# ------------------------------------------------------------------ render entry points
def render_rgb(w, h, spp, max_depth=32, noise_threshold=0.0, min_adaptive_samples=32):
    """optix_render_rgb (src/main.rs:47-67): returns float32 [h, w, 3]."""
    out = np.empty((h, w, 3), np.float32)
    rc = lib().optix_render_rgb(w, h, spp, max_depth, noise_threshold, min_adaptive_samples, _f(out))
    if rc != 0:
        raise RuntimeError(f"optix_render_rgb returned {rc}")
    return out


def render_bayer(w, h, spp, max_depth=32, noise_threshold=0.0, min_adaptive_samples=32, pattern=0):
    """optix_render_bayer_f32 (src/main.rs:81-103): returns float32 [h, w]."""
    out = np.empty((h, w), np.float32)
    rc = lib().optix_render_bayer_f32(w, h, spp, max_depth, noise_threshold, min_adaptive_samples, pattern, _f(out))
    if rc != 0:
        raise RuntimeError(f"optix_render_bayer_f32 returned {rc}")
    return out


def render_accumulate_device(sum_tensor, w, h, sample_begin, sample_count, max_depth, frame):
    """Adds this chunk's per-pixel sums into a CUDA float32 tensor of w*h*3 elements."""
    assert sum_tensor.is_cuda and sum_tensor.is_contiguous() and sum_tensor.numel() == w * h * 3
    rc = lib().pt_render_accumulate_device(w, h, sample_begin, sample_count, max_depth, frame, sum_tensor.data_ptr())
    if rc != 0:
        raise RuntimeError(f"pt_render_accumulate_device returned {rc}")


def render_adaptive_accumulate_device(sum_tensor, count_tensor, w, h, sample_begin, sample_count, max_depth, frame,
                                      noise_threshold, min_adaptive_samples=32):
    """Adaptive share of a sample-sharded render: adds sums (w*h*3) and per-pixel sample counts (w*h) into
    two CUDA float32 tensors."""
    for t, n in ((sum_tensor, w * h * 3), (count_tensor, w * h)):
        assert t.is_cuda and t.dtype.is_floating_point and t.element_size() == 4 and t.numel() == n and t.is_contiguous()
    rc = lib().pt_render_adaptive_accumulate_device(w, h, sample_begin, sample_count, max_depth, frame,
                                                    noise_threshold, min_adaptive_samples, sum_tensor.data_ptr(),
                                                    count_tensor.data_ptr())
    if rc != 0:
        raise RuntimeError(f"pt_render_adaptive_accumulate_device returned {rc}")


def render_adaptive_shared_device(stats_tensor, w, h, sample_begin, sample_count, total_spp, max_depth, frame,
                                  noise_threshold, min_adaptive_samples=32, reduce=None):
    """One rank's share of an adaptive render whose noise test runs on the statistics of all ranks.
    stats_tensor: CUDA float32, w*h*8; reduce(): sums stats_tensor over the ranks in place and leaves the
    result visible to the device (called at every round boundary and once at the end; None: one rank).
    On return stats_tensor.view(h, w, 8) holds (sum rgb, samples, sum of squares rgb, 0) of all ranks."""
    t = stats_tensor
    assert t.is_cuda and t.dtype.is_floating_point and t.element_size() == 4 and t.numel() == w * h * 8 and t.is_contiguous()
    failure = []

    def _call(_user):
        try:
            reduce()
            return 0
        except BaseException as e:  # an exception must not unwind through the C frames
            failure.append(e)
            return 1

    cb = REDUCE_FN(_call) if reduce is not None else REDUCE_FN()
    rc = lib().pt_render_adaptive_shared_device(w, h, sample_begin, sample_count, total_spp, max_depth, frame,
                                                noise_threshold, min_adaptive_samples, t.data_ptr(), cb, None)
    if failure:
        raise failure[0]
    if rc != 0:
        raise RuntimeError(f"pt_render_adaptive_shared_device returned {rc}")


def demosaic(raw, pattern=0, algo=0):
    raw = np.ascontiguousarray(raw, np.float32)
    h, w = raw.shape
    out = np.empty((h, w, 3), np.float32)
    rc = lib().pt_demosaic_f32(_f(raw), w, h, pattern, algo, _f(out))
    if rc != 0:
        raise RuntimeError(f"pt_demosaic_f32 returned {rc}")
    return out


def write_png(path, rgb, exposure=1.0, bits=8):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    return lib().pt_write_png(path.encode(), w, h, _f(rgb), exposure, bits)


def write_exr(path, rgb, exposure=1.0):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    return lib().pt_write_exr(path.encode(), w, h, _f(rgb), exposure)


def read_exr(path):
    w, h = C.c_int(), C.c_int()
    rc = lib().pt_read_exr(path.encode(), C.byref(w), C.byref(h), None, 0)
    if rc != 0:
        raise RuntimeError(f"pt_read_exr header: {rc}")
    out = np.empty((h.value, w.value, 3), np.float32)
    rc = lib().pt_read_exr(path.encode(), C.byref(w), C.byref(h), _f(out), out.size)
    if rc != 0:
        raise RuntimeError(f"pt_read_exr: {rc}")
    return out


def sample_counts(w, h):
    """Per-pixel sample counts of the last render and the number of adaptive rounds it ran."""
    counts = np.zeros((h, w), np.int32)
    rounds = C.c_int(0)
    rc = lib().pt_get_sample_counts(counts.ctypes.data_as(C.POINTER(C.c_int)), C.byref(rounds))
    if rc != 0:
        raise RuntimeError("pt_get_sample_counts: no renderer")
    return counts, rounds.value


def ray_counts():
    a, b = C.c_uint64(), C.c_uint64()
    lib().pt_get_ray_counts(C.byref(a), C.byref(b))
    return a.value, b.value


def kernel_stats(kernel):
    ms, timed, launches, work = C.c_double(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    lib().pt_get_kernel_stats(kernel, C.byref(ms), C.byref(timed), C.byref(launches), C.byref(work))
    return {"total_ms": ms.value, "timed": timed.value, "launches": launches.value, "work": work.value}


def scene_arrays():
    """(vertices [nv,3], indices [nt,3], normals [nt,3], materials [nt,11]) of the current scene; needs no GPU."""
    nv, nt = C.c_int(), C.c_int()
    lib().pt_scene_counts(C.byref(nv), C.byref(nt))
    V = np.zeros((nv.value, 3), np.float32)
    I = np.zeros((nt.value, 3), np.uint32)
    N = np.zeros((nt.value, 3), np.float32)
    M = np.zeros((nt.value, 11), np.float32)
    lib().pt_scene_get(_f(V), I.ctypes.data_as(_u32p), _f(N), _f(M))
    return V, I, N, M


def scene_info():
    nt, nn, dp, nb = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    lib().pt_scene_info(C.byref(nt), C.byref(nn), C.byref(dp), C.byref(nb))
    return {"triangles": nt.value, "nodes": nn.value, "depth": dp.value, "bvh_bytes": nb.value}


def trace_rays(origins, dirs, tmin=None, tmax=None, any_hit=False):
    o = np.ascontiguousarray(origins, np.float32)
    d = np.ascontiguousarray(dirs, np.float32)
    n = o.shape[0]
    t0 = np.zeros(n, np.float32) if tmin is None else np.ascontiguousarray(tmin, np.float32)
    t1 = np.full(n, 1e16, np.float32) if tmax is None else np.ascontiguousarray(tmax, np.float32)
    prim = np.zeros(n, np.int32)
    t = np.zeros(n, np.float32)
    b1 = np.zeros(n, np.float32)
    b2 = np.zeros(n, np.float32)
    lib().pt_trace_rays(n, _f(o), _f(d), _f(t0), _f(t1), 1 if any_hit else 0, _i(prim), _f(t), _f(b1), _f(b2))
    return prim, t, b1, b2


def trace_rays_instances(origins, dirs, tmin=None, tmax=None, any_hit=False):
    """trace_rays that also returns the instance of every closest hit (-1: flat scene or miss)."""
    o = np.ascontiguousarray(origins, np.float32)
    d = np.ascontiguousarray(dirs, np.float32)
    n = o.shape[0]
    t0 = np.zeros(n, np.float32) if tmin is None else np.ascontiguousarray(tmin, np.float32)
    t1 = np.full(n, 1e16, np.float32) if tmax is None else np.ascontiguousarray(tmax, np.float32)
    prim, inst = np.zeros(n, np.int32), np.zeros(n, np.int32)
    t, b1, b2 = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib().pt_trace_rays_instances(n, _f(o), _f(d), _f(t0), _f(t1), 1 if any_hit else 0, _i(prim), _f(t), _f(b1), _f(b2), _i(inst))
    return prim, t, b1, b2, inst


def scene_instanced(meshes, instances):
    """meshes: list of (vertices [nv,3], indices [nt,3], materials [nt,11] or None); instances: list of
    (mesh id, xform [3,4] object -> world, material [11] or None).  Returns the number of instances."""
    L = lib()
    L.pt_scene_instanced_begin()
    for V, I, M in meshes:
        V = np.ascontiguousarray(V, np.float32)
        I = np.ascontiguousarray(I, np.uint32)
        M = None if M is None else np.ascontiguousarray(M, np.float32)
        if L.pt_scene_add_mesh(_f(V), len(V), I.ctypes.data_as(_u32p), len(I), None if M is None else _f(M)) < 0:
            raise ValueError("pt_scene_add_mesh refused the mesh")
    for mesh, xform, mat in instances:
        x = np.ascontiguousarray(xform, np.float32).reshape(12)
        m = None if mat is None else np.ascontiguousarray(mat, np.float32)
        if L.pt_scene_add_instance(mesh, _f(x), None if m is None else _f(m)) < 0:
            raise ValueError("pt_scene_add_instance refused the instance")
    n = L.pt_scene_instanced_commit()
    if n < 0:
        raise ValueError("pt_scene_instanced_commit: no instances")
    return n


def train_samples():
    """The training ring of the last renders as an (n, 10) float32 array: position, dir_in, contrib, and the
    is_delta flag as the bits of column 9 (the TrainSample layout, src/guiding.rs:5-11)."""
    ptr = C.POINTER(TrainSample)()
    cnt = C.c_uint32()
    lib().guiding_map_train_samples(C.byref(ptr), C.byref(cnt))
    if cnt.value == 0:
        return np.zeros((0, 10), np.float32)
    return np.ctypeslib.as_array(C.cast(ptr, _fp), shape=(cnt.value, 10)).copy()


class GuidingField:
    """RAII pair of a trainer device and a field with one reusable sample container (src/guiding.rs:71-143)."""

    def __init__(self, scene_min, scene_max, threads=0):
        L = lib()
        self.dev = L.pgl_dev_create(threads)
        mn, mx = np.asarray(scene_min, np.float32), np.asarray(scene_max, np.float32)
        self.field = L.pgl_field_create(self.dev, _f(mn), _f(mx))
        self.samples = L.pgl_samples_create()

    def update(self, records):
        """records: (n, 10) float32 in the TrainSample layout.  Adds them and updates the field."""
        L = lib()
        records = np.ascontiguousarray(records, np.float32)
        if len(records):
            L.pt_pgl_samples_add_batch(self.samples, records.ctypes.data_as(C.c_void_p), len(records))
        L.pgl_field_update(self.field, self.samples)

    BIN_LOBES, BIN_WORDS, BIN_SHIFT = 6, 8, 24

    def layout(self):
        """(bmin[3], bmax[3], res[3], axes (regions, 6, 3) float32): what assigning a sample to a region and lobe needs."""
        L = lib()
        bmin, bmax, res, n = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_int * 3)(), C.c_uint32()
        if L.pt_pgl_field_layout(self.field, bmin, bmax, res, None, C.byref(n)) != 0:
            raise RuntimeError("pt_pgl_field_layout failed")
        axes = np.zeros((n.value, self.BIN_LOBES, 3), np.float32)
        L.pt_pgl_field_layout(self.field, None, None, None, _f(axes), None)
        return np.array(bmin[:], np.float32), np.array(bmax[:], np.float32), np.array(res[:], np.int32), axes

    def update_binned(self, bins):
        """bins: int64 (regions * 6 * 8) fixed-point sums per region and lobe (pt_guiding_bin_train_samples_device)."""
        bins = np.ascontiguousarray(bins, np.int64)
        lib().pt_pgl_field_update_binned(self.field, bins.ctypes.data_as(C.c_void_p))

    def bin_ring_device(self, d_bins_ptr):
        """Adds the current renderer's training ring into the int64 table at device address d_bins_ptr."""
        if lib().pt_guiding_bin_train_samples_device(self.field, C.c_void_p(d_bins_ptr)) != 0:
            raise RuntimeError("pt_guiding_bin_train_samples_device failed")

    def update_from_ring(self):
        """Bin on the GPU, copy the table, update, empty the ring: the whole of src/guiding.rs:194-236 in one call."""
        if lib().pt_pgl_field_update_from_ring(self.field) != 0:
            raise RuntimeError("pt_pgl_field_update_from_ring failed")

    def snapshot(self):
        """(regions (n, 8) float32, lobes (m, 8) float32) in the layout upload_snapshot takes; the two integer
        fields of a region are stored as float values."""
        L = lib()
        regs, lobes = C.POINTER(PglRegion)(), C.POINTER(PglLobe)()
        nr, nl = C.c_uint32(), C.c_uint32()
        L.pgl_field_snapshot(self.field, C.byref(regs), C.byref(nr), C.byref(lobes), C.byref(nl))
        R = np.zeros((nr.value, 8), np.float32)
        for i in range(nr.value):
            R[i, 0:3], R[i, 3:6], R[i, 6], R[i, 7] = regs[i].bmin[:], regs[i].bmax[:], regs[i].lobe_ofs, regs[i].lobe_num
        Lb = np.zeros((nl.value, 8), np.float32)
        for i in range(nl.value):
            Lb[i, 0:3], Lb[i, 3], Lb[i, 4], Lb[i, 5:8] = lobes[i].mu[:], lobes[i].kappa, lobes[i].weight, lobes[i].rgb[:]
        return R, Lb

    def close(self):
        L = lib()
        if self.samples:
            L.pgl_samples_destroy(self.samples)
            L.pgl_field_destroy(self.field)
            L.pgl_dev_destroy(self.dev)
            self.samples = self.field = self.dev = None


def upload_snapshot(regions, lobes):
    """regions: (n,8) [bmin3,bmax3,ofs,num]; lobes: (m,8) [mu3,kappa,weight,rgb3]."""
    n, m = len(regions), len(lobes)
    R = (PglRegion * max(n, 1))()
    for i, r in enumerate(regions):
        R[i].bmin[:] = [float(x) for x in r[0:3]]
        R[i].bmax[:] = [float(x) for x in r[3:6]]
        R[i].lobe_ofs = int(r[6])
        R[i].lobe_num = int(r[7])
    Lb = (PglLobe * max(m, 1))()
    for i, l in enumerate(lobes):
        Lb[i].mu[:] = [float(x) for x in l[0:3]]
        Lb[i].kappa = float(l[3])
        Lb[i].weight = float(l[4])
        Lb[i].rgb[:] = [float(x) for x in l[5:8]]
    lib().guiding_upload_snapshot(R, n, Lb, m)
