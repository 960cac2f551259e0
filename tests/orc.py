"""ctypes binding of the CPU oracle (oracle/liborc.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liborc.so")

STRICT, CONVERGE = 0, 1


class Material(C.Structure):
    _fields_ = [("base", C.c_float * 3), ("metallic", C.c_float), ("roughness", C.c_float),
                ("ior", C.c_float), ("alpha", C.c_float), ("transmission", C.c_float),
                ("emission", C.c_float * 3)]


class Region(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("lobe_ofs", C.c_uint32), ("lobe_num", C.c_uint32)]


class Lobe(C.Structure):
    _fields_ = [("mu", C.c_float * 3), ("kappa", C.c_float), ("weight", C.c_float), ("rgb", C.c_float * 3)]


class TrainSample(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("dir_in", C.c_float * 3), ("contrib", C.c_float * 3),
                ("is_delta", C.c_uint32)]


class Guide(C.Structure):
    _fields_ = [("regions", C.POINTER(Region)), ("region_count", C.c_uint32),
                ("lobes", C.POINTER(Lobe)), ("lobe_count", C.c_uint32),
                ("grid", C.POINTER(C.c_uint32)),
                ("grid_res", C.c_int * 3), ("grid_min", C.c_float * 3), ("grid_max", C.c_float * 3),
                ("cell_size", C.c_float * 3)]


class PRD(C.Structure):
    _fields_ = [("radiance", C.c_float * 3), ("throughput", C.c_float * 3), ("origin", C.c_float * 3),
                ("direction", C.c_float * 3), ("seed", C.c_uint64), ("depth", C.c_int), ("done", C.c_int),
                ("prev_pdf_bsdf", C.c_float), ("prev_pdf_valid", C.c_int), ("px", C.c_int), ("py", C.c_int),
                ("sample", C.c_int), ("rng_dim", C.c_int)]


class Hit(C.Structure):
    _fields_ = [("prim", C.c_int), ("t", C.c_float), ("b1", C.c_float), ("b2", C.c_float)]


class RenderArgs(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("spp", C.c_int), ("max_depth", C.c_int),
                ("frame", C.c_int), ("bayer_pattern", C.c_int), ("noise_threshold", C.c_float),
                ("min_adaptive_samples", C.c_int), ("mode", C.c_int), ("sample_begin", C.c_int),
                ("adaptive_deterministic", C.c_int), ("nthreads", C.c_int), ("adaptive_round", C.c_int),
                ("guide", C.POINTER(Guide)), ("train", C.POINTER(TrainSample)),
                ("train_capacity", C.c_uint32), ("train_write_idx", C.POINTER(C.c_uint32))]


def build(force=False):
    """Compile the oracle in place (gcc); the GPU box uses the prebuilt file."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("pt_oracle.c", "demosaic_oracle.c", "pt_oracle.h", "Makefile")]
    if not force and os.path.exists(LIB_PATH) and all(
            os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    fp, ip, u32p = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint32)
    L.orc_pcg.restype = C.c_uint32
    L.orc_pcg.argtypes = [C.POINTER(C.c_uint64)]
    L.orc_rnd.restype = C.c_float
    L.orc_rnd.argtypes = [C.POINTER(C.c_uint64)]
    L.orc_hash_u32.restype = C.c_uint32
    L.orc_hash_u32.argtypes = [C.c_uint32]
    L.orc_blue_noise.argtypes = [C.c_int] * 4 + [fp]
    L.orc_sample_tri.restype = C.c_float
    L.orc_sample_tri.argtypes = [C.c_float]
    L.orc_olpf_jitter.argtypes = [C.c_int] * 4 + [fp]
    L.orc_pixel_seed.restype = C.c_uint64
    L.orc_pixel_seed.argtypes = [C.c_int, C.c_uint32]
    L.orc_next_rand.restype = C.c_float
    L.orc_next_rand.argtypes = [C.POINTER(PRD), C.c_int]
    L.orc_make_onb.argtypes = [fp, fp, fp]
    L.orc_sample_ggx_vndf.argtypes = [fp, C.c_float, fp, fp, fp, fp, fp]
    L.orc_camera_default.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp]
    L.orc_camera_dir.argtypes = [C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, fp]
    L.orc_bayer_channel_for.restype = C.c_int
    L.orc_bayer_channel_for.argtypes = [C.c_int] * 3
    L.orc_scene_num_triangles.restype = C.c_int
    L.orc_scene_num_vertices.restype = C.c_int
    L.orc_scene_vertices.restype = fp
    L.orc_scene_indices.restype = u32p
    L.orc_scene_normals.restype = fp
    L.orc_scene_materials.restype = C.POINTER(Material)
    L.orc_scene_set.argtypes = [fp, C.c_int, u32p, C.c_int, C.POINTER(Material)]
    L.orc_trace_closest.restype = Hit
    L.orc_trace_closest.argtypes = [fp, fp, C.c_float, C.c_float]
    L.orc_trace_any.restype = C.c_int
    L.orc_trace_any.argtypes = [fp, fp, C.c_float, C.c_float]
    L.orc_closesthit.argtypes = [C.POINTER(PRD), C.POINTER(Hit), C.c_int, C.c_int, C.POINTER(Guide),
                                 C.POINTER(C.c_uint64)]
    L.orc_guiding_region_id.restype = C.c_int
    L.orc_guiding_region_id.argtypes = [C.POINTER(Guide), fp]
    L.orc_vmf_sample.argtypes = [fp, C.c_float, fp, fp]
    L.orc_vmf_pdf.restype = C.c_float
    L.orc_vmf_pdf.argtypes = [fp, C.c_float, fp]
    L.orc_guiding_choose_lobe.restype = C.c_int
    L.orc_guiding_choose_lobe.argtypes = [C.POINTER(Region), C.POINTER(Lobe), C.c_float, C.c_int]
    L.orc_guiding_mixture_pdf.restype = C.c_float
    L.orc_guiding_mixture_pdf.argtypes = [C.POINTER(Region), C.POINTER(Lobe), fp]
    L.orc_guiding_build_grid.argtypes = [C.POINTER(Region), C.c_uint32, ip, u32p, ip, fp, fp, fp]
    L.orc_render.restype = C.c_int
    L.orc_render.argtypes = [C.POINTER(RenderArgs), fp, fp, ip, C.POINTER(C.c_uint64)]
    L.orc_set_on_plane_flags.argtypes = [C.c_void_p]
    L.orc_set_roulette.argtypes = [C.c_int]
    L.orc_cfa_at.restype = C.c_int
    L.orc_cfa_at.argtypes = [C.c_int] * 3
    L.orc_demosaic_amaze.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp]
    L.orc_demosaic_bilinear.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp]
    _lib = L
    return L


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def make_guide(regions, lobes, grid_res):
    """regions: (n,8) float array [bmin3,bmax3,ofs,num]; lobes: (m,8) [mu3,kappa,weight,rgb3].

    Returns (Guide, keepalive) with the grid built by the oracle's restatement of
    optix_wrapper.cpp:1191-1242."""
    L = lib()
    n, m = len(regions), len(lobes)
    R = (Region * max(n, 1))()
    for i, r in enumerate(regions):
        R[i].bmin[:] = [float(x) for x in r[0:3]]
        R[i].bmax[:] = [float(x) for x in r[3:6]]
        R[i].lobe_ofs = int(r[6])
        R[i].lobe_num = int(r[7])
    Lb = (Lobe * max(m, 1))()
    for i, l in enumerate(lobes):
        Lb[i].mu[:] = [float(x) for x in l[0:3]]
        Lb[i].kappa = float(l[3])
        Lb[i].weight = float(l[4])
        Lb[i].rgb[:] = [float(x) for x in l[5:8]]
    res_in = (C.c_int * 3)(*[int(x) for x in grid_res])
    res = [max(1, int(x)) for x in grid_res]
    grid = (C.c_uint32 * (res[0] * res[1] * res[2]))()
    G = Guide()
    L.orc_guiding_build_grid(R, n, res_in, grid, G.grid_res, G.grid_min, G.grid_max, G.cell_size)
    G.regions = C.cast(R, C.POINTER(Region))
    G.region_count = n
    G.lobes = C.cast(Lb, C.POINTER(Lobe))
    G.lobe_count = m
    G.grid = C.cast(grid, C.POINTER(C.c_uint32))
    return G, (R, Lb, grid)


def render(width, height, spp, max_depth=32, frame=1, pattern=0, noise_threshold=0.0, min_adaptive=32,
           mode=CONVERGE, sample_begin=0, adaptive_deterministic=0, nthreads=0, guide=None, adaptive_round=0,
           train=None, on_plane=None):
    """Returns (rgb[h,w,3], bayer[h,w], counts[h,w], (radiance_rays, shadow_rays)).
    on_plane: a list; appends a uint8 [h,w] map of the pixels one of whose rays started in the plane of a triangle.
    train: a list; the render then runs on one thread with a training buffer (optix_device.cu:431-441) and
    appends one (n, 9) float32 array -- position, dir_in, contrib of every captured vertex, in capture order."""
    L = lib()
    if nthreads <= 0:
        nthreads = os.cpu_count() or 1
    tbuf, tidx, tcap = None, None, 0
    if train is not None:
        tcap = 1 << 20
        tbuf = (TrainSample * tcap)()
        tidx = C.c_uint32(0)
    a = RenderArgs(width, height, spp, max_depth, frame, pattern, noise_threshold, min_adaptive, mode,
                   sample_begin, adaptive_deterministic, nthreads, adaptive_round,
                   C.pointer(guide) if guide is not None else None, tbuf, tcap, C.pointer(tidx) if tidx is not None else None)
    rgb = np.zeros((height, width, 3), np.float32)
    bayer = np.zeros((height, width), np.float32)
    counts = np.zeros((height, width), np.int32)
    rays = (C.c_uint64 * 2)()
    if on_plane is not None:
        flags = np.zeros((height, width), np.uint8)
        L.orc_set_on_plane_flags(flags.ctypes.data_as(C.c_void_p))
    rc = L.orc_render(C.byref(a), fptr(rgb), fptr(bayer), counts.ctypes.data_as(C.POINTER(C.c_int)), rays)
    if on_plane is not None:
        L.orc_set_on_plane_flags(None)
        on_plane.append(flags)
    assert rc == 0
    if train is not None:
        assert tidx.value <= tcap, "more vertices than the test buffer holds"
        rec = np.frombuffer(tbuf, dtype=np.float32).reshape(tcap, 10)[: tidx.value, :9].copy()
        train.append(rec)
    return rgb, bayer, counts, (int(rays[0]), int(rays[1]))


def demosaic(raw, pattern, algo="amaze"):
    L = lib()
    raw = np.ascontiguousarray(raw, np.float32)
    h, w = raw.shape
    out = np.zeros((h, w, 3), np.float32)
    fn = L.orc_demosaic_amaze if algo == "amaze" else L.orc_demosaic_bilinear
    fn(fptr(raw), w, h, pattern, fptr(out))
    return out
