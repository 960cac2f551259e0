"""CPU tests of everything around the kernels: the C-ABI surface, the host BVH builder, the
device headers compiled for the host (tests/hostcheck) against the oracle, EXR I/O, the
stand-in guiding trainer, and the multi-process reduce path (gloo, world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def P(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


# ------------------------------------------------------------------ C-ABI surface
def test_library_exports_every_declared_symbol(pt):
    """Every PT_EXPORT declaration in include/pt_api.h resolves in libpt_hip.so and is bound."""
    hdr = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    declared = set(re.findall(r"PT_EXPORT\s+[\w\s\*]+?\b(\w+)\s*\(", hdr))
    assert len(declared) >= 50
    L = pt.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared but not exported"
    assert declared == set(pt.SIGNATURES), declared ^ set(pt.SIGNATURES)
    out = subprocess.check_output(["nm", "-D", "--defined-only", pt.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert declared <= exported
    # the reference's exported names (src/main.rs:9-32, src/guiding.rs:31-68) are all there
    for name in ["optix_render_rgb", "optix_render_bayer_f32", "optix_alloc_host", "optix_free_host",
                 "optix_synchronize", "pgl_dev_create", "pgl_field_snapshot", "guiding_upload_snapshot",
                 "guiding_map_train_samples", "guiding_set_grid_res"]:
        assert name in exported


def test_product_has_no_oracle_or_cpu_fallback(pt):
    """The shipped library and package never reference the oracle."""
    out = subprocess.check_output(["nm", "-D", pt.LIB_PATH], text=True)
    assert "orc_" not in out and "hc_" not in out
    for dirpath, _, files in os.walk(pt.ROOT):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(dirpath, f)).read()
                assert "liborc" not in src and "pt_oracle" not in src and "hostcheck" not in src or f == "pt_types.h", f


def test_struct_layouts_match_reference_sizes(pt):
    # sizes in the reference: pgl_region 32 B, pgl_lobe 32 B, TrainSample 40 B (src/guiding.rs:5-28)
    assert C.sizeof(pt.PglRegion) == 32 and C.sizeof(pt.PglLobe) == 32 and C.sizeof(pt.TrainSample) == 40


# ------------------------------------------------------------------ BVH builder + traversal logic
def _scene_arrays(hc):
    nt, nn, dp = C.c_int(), C.c_int(), C.c_int()
    hc.hc_scene_info(C.byref(nt), C.byref(nn), C.byref(dp))
    nv = hc.hc_num_vertices()
    V = np.zeros((nv, 3), np.float32)
    I = np.zeros((nt.value, 3), np.uint32)
    N = np.zeros((nt.value, 3), np.float32)
    M = np.zeros((nt.value, 11), np.float32)
    hc.hc_scene_get(P(V), P(I, C.c_uint32), P(N), P(M))
    return V, I, N, M, nn.value, dp.value


def test_builtin_scene_equals_oracle_scene(hostcheck, oracle):
    hostcheck.hc_scene_cornell()
    V, I, N, M, nodes, depth = _scene_arrays(hostcheck)
    L = oracle.lib()
    assert np.array_equal(V, np.ctypeslib.as_array(L.orc_scene_vertices(), shape=(48, 3)))
    assert np.array_equal(I, np.ctypeslib.as_array(L.orc_scene_indices(), shape=(24, 3)))
    assert np.array_equal(N, np.ctypeslib.as_array(L.orc_scene_normals(), shape=(24, 3)))
    OM = L.orc_scene_materials()
    for i in range(24):
        want = list(OM[i].base) + [OM[i].metallic, OM[i].roughness, OM[i].ior, OM[i].alpha, OM[i].transmission] + list(OM[i].emission)
        assert M[i].tolist() == pytest.approx(want, abs=0)
    assert depth <= 3 and nodes <= 8


def _random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.99, 0.99, (n, 3)).astype(np.float32)
    o[:, 2] = rng.uniform(0.01, 1.99, n)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def _hc_trace(hc, o, d, tmax=None, any_hit=0):
    n = len(o)
    tmin = np.zeros(n, np.float32)
    tmax = np.full(n, 1e16, np.float32) if tmax is None else tmax
    prim = np.zeros(n, np.int32)
    t, b1, b2 = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    hc.hc_trace_rays(n, P(o), P(d), P(tmin), P(tmax), any_hit, P(prim, C.c_int), P(t), P(b1), P(b2))
    return prim, t, b1, b2


def _orc_trace(oracle, o, d):
    L = oracle.lib()
    fp = C.POINTER(C.c_float)
    prim = np.zeros(len(o), np.int32)
    t = np.zeros(len(o), np.float32)
    for i in range(len(o)):
        h = L.orc_trace_closest(o[i].ctypes.data_as(fp), d[i].ctypes.data_as(fp), 0.0, 1e16)
        prim[i], t[i] = h.prim, h.t
    return prim, t


def test_traversal_logic_matches_brute_force_cornell(hostcheck, oracle):
    hostcheck.hc_scene_cornell()
    o, d = _random_rays(20000, 1)
    prim, t, b1, b2 = _hc_trace(hostcheck, o, d)
    op, ot = _orc_trace(oracle, o, d)
    assert (prim != op).sum() == 0
    hit = op >= 0
    assert np.abs(t[hit] - ot[hit]).max() < 1e-5
    assert ((b1 >= 0) & (b2 >= 0) & (b1 + b2 <= 1.0 + 1e-5))[hit].all()


def test_traversal_logic_matches_brute_force_soup(hostcheck, oracle):
    """A deeper tree: 64 spheres x 320 triangles + the room; rays checked against the oracle's
    brute-force loop over the same triangles."""
    hostcheck.hc_scene_synthetic(7, 64, 2)
    try:
        V, I, N, M, nodes, depth = _scene_arrays(hostcheck)
        assert len(I) == 12 + 64 * 320 and depth >= 4
        oracle.lib().orc_scene_set(P(V), len(V), P(I, C.c_uint32), len(I), None)
        o, d = _random_rays(3000, 2)
        prim, t, _, _ = _hc_trace(hostcheck, o, d)
        op, ot = _orc_trace(oracle, o, d)
        assert (prim != op).sum() == 0
        assert np.abs(t - ot)[op >= 0].max() < 1e-5
        tmax = np.random.default_rng(3).uniform(0.05, 2.0, len(o)).astype(np.float32)
        occ, _, _, _ = _hc_trace(hostcheck, o, d, tmax, any_hit=1)
        L = oracle.lib()
        fp = C.POINTER(C.c_float)
        want = np.array([L.orc_trace_any(o[i].ctypes.data_as(fp), d[i].ctypes.data_as(fp), 0.0, float(tmax[i]))
                         for i in range(len(o))])
        assert np.array_equal(occ, want)
    finally:
        oracle.lib().orc_scene_set(None, 0, None, 0, None)
        hostcheck.hc_scene_cornell()


def test_awkward_geometry_matches_brute_force(hostcheck, oracle):
    """Degenerate triangles, slivers, tiny triangles far from the origin: the fast triangle test's error
    band has to hand every doubtful case to the watertight test, so the answers stay sound against a float64
    ground truth (and so do the oracle's)."""
    import geom_cases
    V, I, centre, rng = geom_cases.awkward_scene()
    hostcheck.hc_scene_triangles(P(V), len(V), P(I, C.c_uint32), len(I), None)
    oracle.lib().orc_scene_set(P(V), len(V), P(I, C.c_uint32), len(I), None)
    try:
        o, d = geom_cases.awkward_rays(V, I, centre, rng)
        prim, t, b1, b2 = _hc_trace(hostcheck, o, d)
        definite_miss, strict_miss, t_first, rows = geom_cases.check_against_float64(V, I, o, d, prim, t, b1, b2)
        # the oracle's brute-force loop passes the same checks on the same rays
        op, ot = _orc_trace(oracle, o, d)
        oh = op >= 0
        assert not strict_miss[rows[oh], op[oh]].any() and not np.isfinite(t_first[~oh]).any()
        assert (prim != op).mean() < 0.08  # they differ only inside the margins
    finally:
        oracle.lib().orc_scene_set(None, 0, None, 0, None)
        hostcheck.hc_scene_cornell()


def test_parallelogram_items_give_the_triangles_answers(hostcheck):
    """Two consecutive triangles that form a parallelogram share one traversal record; the partner is found
    through the mirrored barycentrics of the first.  Every diagonal / winding / corner-order combination must
    report the right primitive with that primitive's own barycentrics, look-alikes must not be paired, and
    the answers must be those of the tree built without pairing."""
    import geom_cases
    V, I, centre, rng, n_quads = geom_cases.quad_scene()
    hostcheck.hc_scene_triangles(P(V), len(V), P(I, C.c_uint32), len(I), None)
    try:
        assert hostcheck.hc_num_items() == len(I) - n_quads  # every true parallelogram, nothing else
        o, d = geom_cases.quad_rays(V, I, rng)
        prim, t, b1, b2 = _hc_trace(hostcheck, o, d)
        geom_cases.check_against_float64(V, I, o, d, prim, t, b1, b2)
        assert (prim[prim >= 0] % 2 == 1).mean() > 0.3  # second halves are reported as themselves
        tmax = rng.uniform(0.2, 4.0, len(o)).astype(np.float32)
        occ, _, _, _ = _hc_trace(hostcheck, o, d, tmax, any_hit=1)
        hostcheck.hc_set_pairing(0)
        assert hostcheck.hc_num_items() == len(I)
        prim0, t0, b10, b20 = _hc_trace(hostcheck, o, d)
        occ0, _, _, _ = _hc_trace(hostcheck, o, d, tmax, any_hit=1)
        same = prim == prim0
        assert same.mean() > 0.98  # rays aimed exactly at shared edges may report either neighbour
        hit = same & (prim >= 0)
        db = np.maximum(np.abs(b1[hit] - b10[hit]), np.abs(b2[hit] - b20[hit]))
        assert np.abs(t[hit] - t0[hit]).max() < 2e-3 and db.max() < 5e-3  # grazing rays 40 units out
        assert np.quantile(np.abs(t[hit] - t0[hit]), 0.999) < 2e-4 and np.quantile(db, 0.999) < 2e-4
        assert ((occ >= 0) == (occ0 >= 0)).mean() > 0.999
    finally:
        hostcheck.hc_set_pairing(-1)
        hostcheck.hc_scene_cornell()
    assert hostcheck.hc_num_items() == 12  # the built-in scene is twelve quads


def test_wide_nodes_give_the_packed_trees_hits(hostcheck):
    """Small trees also exist as wide nodes for the kernels that keep the scene in LDS: unquantised boxes in
    world coordinates, up to 16 slots, subtrees of at most 16 leaves flattened into one node (the built-in
    scene: one node).  Fewer candidates and node visits, but every hit (primitive, distance, barycentrics,
    occlusion) is the packed tree's, bit for bit -- including rays aimed at edges and corners of the closed box
    -- and so is a whole render with the origin-plane skip, which uses the wide tree's own home table."""
    import geom_cases
    rng = np.random.default_rng(11)

    def both(o, d, want_exact=True):
        assert hostcheck.hc_has_wide_nodes() == (1 if want_exact else 0)
        out = []
        tmax = rng.uniform(0.2, 4.0, len(o)).astype(np.float32)
        for mode in (0, 1):
            hostcheck.hc_set_wide_nodes(mode)
            out.append(_hc_trace(hostcheck, o, d) + (_hc_trace(hostcheck, o, d, tmax, any_hit=1)[0],))
        hostcheck.hc_set_wide_nodes(0)
        for a, b in zip(*out):
            assert np.array_equal(a, b)

    try:
        hostcheck.hc_scene_cornell()
        assert hostcheck.hc_wide_depth() == 1  # twelve quads: one flat node
        both(*_random_rays(40000, 6))
        # rays from inside the room at points on edges, corners and quad diagonals
        targets = []
        for s in np.linspace(-1, 0.95, 60):
            targets += [(s, s, 0.0), (s, -1.0, 1.0 + s), (-1.0, s, 1.0 - s), (1.0, s, 1.0 + s), (s, -s, 2.0),
                        (s, -1.0, 0.0), (-1.0, -1.0, 1.0 + s), (1.0, -1.0, 1.0 + s), (s, s, 2.0)]
        targets = np.array(targets, np.float32)
        org = np.tile(np.array([[0.123, 0.3, 1.1]], np.float32), (len(targets), 1))
        d = targets - org
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        both(org, d.astype(np.float32))
        prim = _hc_trace(hostcheck, org, d.astype(np.float32))[0]
        assert (prim >= 0).all()
        hostcheck.hc_scene_synthetic(7, 2, 1)  # a few levels, still small enough to keep the bounds
        o = rng.uniform(-0.9, 0.9, (20000, 3)).astype(np.float32)
        o[:, 2] += 1.0
        d = rng.normal(size=(20000, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        both(o, d)
        V, I, centre, qrng, n_quads = geom_cases.quad_scene()
        hostcheck.hc_scene_triangles(P(V), len(V), P(I, C.c_uint32), len(I), None)
        both(*geom_cases.quad_rays(V, I, qrng), want_exact=bool(hostcheck.hc_has_wide_nodes()))
        # whole renders: packed tree, wide tree, wide tree without the origin-plane skip -- the same bits
        hostcheck.hc_scene_cornell()
        w, h = 40, 30
        outs = []
        for wide, skip in ((0, 1), (1, 1), (1, 0)):
            hostcheck.hc_set_wide_nodes(wide)
            hostcheck.hc_set_origin_skip(skip)
            out = np.zeros((h, w, 3), np.float32)
            rays = (C.c_ulonglong * 2)()
            hostcheck.hc_render(w, h, 0, 8, 32, 1, 0, 4, P(out), rays)
            outs.append((out, rays[0], rays[1]))
        for o in outs[1:]:
            assert np.array_equal(outs[0][0], o[0]) and outs[0][1:] == o[1:]
    finally:
        hostcheck.hc_set_origin_skip(1)
        hostcheck.hc_set_wide_nodes(0)
        hostcheck.hc_scene_cornell()


def test_watertight_no_leak_through_shared_edges(hostcheck):
    """Rays aimed exactly at quad diagonals, edges and corners of the closed part of the box
    must hit something (the two triangles of a quad share their diagonal)."""
    hostcheck.hc_scene_cornell()
    o = np.tile(np.array([[0.123, 0.456, 1.0]], np.float32), (0, 1))
    targets = []
    for s in np.linspace(-1, 0.95, 40):  # stay off the rim of the open front (y = +1)
        targets += [(s, s * 1.0, 0.0), (s, -1.0, 1.0 + s), (-1.0, s, 1.0 - s), (1.0, s, 1.0 + s), (s, -s, 2.0),
                    (s, -1.0, 0.0), (-1.0, -1.0, 1.0 + s), (1.0, -1.0, 1.0 + s), (s, s, 2.0)]
    targets = np.array(targets, np.float32)
    org = np.tile(np.array([[0.123, 0.3, 1.1]], np.float32), (len(targets), 1))
    d = targets - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    prim, t, _, _ = _hc_trace(hostcheck, org, d.astype(np.float32))
    assert (prim >= 0).all()


def test_empty_and_tiny_scenes(hostcheck):
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    idx = np.array([[0, 1, 2]], np.uint32)
    hostcheck.hc_scene_triangles(P(tri), 3, P(idx, C.c_uint32), 1, None)
    try:
        o = np.array([[0.2, 0.2, 1.0], [2.0, 2.0, 1.0], [0.2, 0.2, -1.0]], np.float32)
        d = np.array([[0, 0, -1], [0, 0, -1], [0, 0, 1]], np.float32)
        prim, t, b1, b2 = _hc_trace(hostcheck, o, d)
        assert prim.tolist() == [0, -1, 0]  # two-sided
        assert t[0] == pytest.approx(1.0) and b1[0] == pytest.approx(0.2) and b2[0] == pytest.approx(0.2)
        hostcheck.hc_scene_triangles(P(tri), 3, P(idx, C.c_uint32), 0, None)
        prim, _, _, _ = _hc_trace(hostcheck, o, d)
        assert (prim == -1).all()
    finally:
        hostcheck.hc_scene_cornell()


# ------------------------------------------------------------------ shading logic vs oracle
class HcRec(C.Structure):
    _fields_ = [("radiance", C.c_float * 3), ("throughput", C.c_float * 3), ("origin", C.c_float * 3),
                ("direction", C.c_float * 3), ("seed", C.c_uint64), ("depth", C.c_int), ("done", C.c_int),
                ("prev_pdf_bsdf", C.c_float), ("prev_pdf_valid", C.c_int), ("px", C.c_int), ("py", C.c_int),
                ("sample", C.c_int), ("rng_dim", C.c_int)]


def make_shade_cases(oracle, n, seed, max_depth=32):
    """Random path states arriving at random primitives of the built-in scene."""
    L = oracle.lib()
    rng = np.random.default_rng(seed)
    V = np.ctypeslib.as_array(L.orc_scene_vertices(), shape=(48, 3))
    I = np.ctypeslib.as_array(L.orc_scene_indices(), shape=(24, 3))
    cases = []
    for _ in range(n):
        prim = int(rng.integers(0, 24))
        b1, b2 = rng.uniform(0.02, 0.45, 2)
        Pt = V[I[prim, 0]] * (1 - b1 - b2) + V[I[prim, 1]] * b1 + V[I[prim, 2]] * b2
        org = rng.uniform(-0.9, 0.9, 3)
        org[2] = rng.uniform(0.1, 1.9)
        d = Pt - org
        d /= np.linalg.norm(d)
        r = oracle.PRD()
        r.throughput[:] = rng.uniform(0.1, 1.0, 3).tolist()
        r.origin[:] = org.tolist()
        r.direction[:] = d.tolist()
        r.seed = int(rng.integers(0, 2 ** 63))
        r.depth = int(rng.choice([0, 1, 5, max_depth - 3, max_depth - 2, max_depth - 1]))
        r.prev_pdf_bsdf = float(rng.uniform(0.05, 3.0))
        r.prev_pdf_valid = int(rng.integers(0, 2))
        r.px, r.py, r.sample = int(rng.integers(0, 1920)), int(rng.integers(0, 1080)), int(rng.integers(0, 1024))
        r.rng_dim = int(rng.choice([0, 1, 2, 7]))
        cases.append((r, prim, float(np.float32(b1)), float(np.float32(b2))))
    return cases


def check_shade_against_oracle(oracle, cases, shade_fn, frame=1, max_depth=32, guide=None, rtol=2e-3):
    """shade_fn(recs, prims, b1, b2) -> (recs_out, shadow flags).  Returns the fraction of cases
    that agree with the oracle's closest-hit."""
    L = oracle.lib()
    n = len(cases)
    recs = (HcRec * n)()
    prims = np.zeros(n, np.int32)
    b1 = np.zeros(n, np.float32)
    b2 = np.zeros(n, np.float32)
    for i, (r, p, u, v) in enumerate(cases):
        C.memmove(C.byref(recs[i]), C.byref(r), C.sizeof(HcRec))
        prims[i], b1[i], b2[i] = p, u, v
    out, shadow = shade_fn(recs, prims, b1, b2)
    agree = 0
    for i, (r0, p, u, v) in enumerate(cases):
        a = oracle.PRD.from_buffer_copy(r0)
        a.radiance[:] = [0, 0, 0]
        h = oracle.Hit(p, 0.0, u, v)
        sh = C.c_uint64(0)
        L.orc_closesthit(C.byref(a), C.byref(h), frame, max_depth, C.byref(guide) if guide is not None else None, C.byref(sh))
        g = out[i]
        ca = np.array(list(a.radiance)) * np.array(list(a.throughput))
        cg = np.array(list(g.radiance))
        ok = (a.seed == g.seed and a.done == g.done and a.depth == g.depth and int(sh.value) == int(shadow[i])
              and a.prev_pdf_valid == g.prev_pdf_valid
              and np.allclose(ca, cg, rtol=rtol, atol=1e-5)
              and np.allclose(list(a.throughput), list(g.throughput), rtol=rtol, atol=1e-6)
              and (a.done or (np.allclose(list(a.direction), list(g.direction), atol=2e-4)
                              and np.allclose(list(a.origin), list(g.origin), atol=1e-5)))
              and abs(a.prev_pdf_bsdf - g.prev_pdf_bsdf) <= rtol * abs(a.prev_pdf_bsdf) + 1e-6)
        agree += bool(ok)
    return agree / n


def test_shading_logic_matches_oracle(hostcheck, oracle):
    hostcheck.hc_scene_cornell()
    cases = make_shade_cases(oracle, 4000, 21)

    def shade(recs, prims, b1, b2):
        shadow = np.zeros(len(prims), np.int32)
        hostcheck.hc_shade(len(prims), recs, P(prims, C.c_int), P(b1), P(b2), 1, 32, P(shadow, C.c_int))
        return recs, shadow

    frac = check_shade_against_oracle(oracle, cases, shade)
    assert frac > 0.995, frac


def test_render_logic_strict_matches_oracle(hostcheck, oracle):
    """Same seeds, same paths: the per-pixel sample loop of the kernels reproduces the oracle's
    4-sample launch except where a grazing hit flips (a path leaving the glass box through its
    bottom face starts at z ~ 0 and may or may not see the floor)."""
    hostcheck.hc_scene_cornell()
    w, h = 48, 36
    out = np.zeros((h, w, 3), np.float32)
    rays = (C.c_ulonglong * 2)()
    hostcheck.hc_render(w, h, 0, 4, 32, 1, 1, 4, P(out), rays)
    ref, _, _, orays = oracle.render(w, h, 4, frame=1, mode=oracle.STRICT)
    close = (np.abs(out / 4 - ref) <= 1e-3 + 1e-3 * np.abs(ref)).all(axis=2).mean()
    assert close > 0.93, close
    # a handful of 32-bounce paths inside the glass box flip with the last bit of a hit distance
    assert abs(rays[0] - orays[0]) / orays[0] < 0.02 and abs(rays[1] - orays[1]) / orays[1] < 0.02


def test_bvh_build_does_not_depend_on_thread_count(hostcheck):
    """Subtrees are built, collapsed and emitted by several threads into a layout fixed up front."""
    hostcheck.hc_bvh_hash.restype = C.c_ulonglong
    hashes = []
    for threads in ("1", "3", None):
        if threads is None:
            os.environ.pop("PT_BVH_THREADS", None)
        else:
            os.environ["PT_BVH_THREADS"] = threads
        hostcheck.hc_scene_synthetic(7, 64, 3)  # 82k triangles: above the size where threads are spawned
        hashes.append(hostcheck.hc_bvh_hash())
    os.environ.pop("PT_BVH_THREADS", None)
    hostcheck.hc_scene_cornell()
    assert hashes[0] == hashes[1] == hashes[2]


def test_skipping_the_origin_plane_changes_nothing(hostcheck):
    """trace_ray may be told which primitive a ray leaves and then skips the triangles in that plane; the
    image and the ray counts must be the same bits as without the hint (Cornell box and a scene of spheres)."""
    for scene in ("cornell", "spheres"):
        if scene == "cornell":
            hostcheck.hc_scene_cornell()
        else:
            hostcheck.hc_scene_synthetic(7, 27, 2)
        w, h = 40, 30
        outs = []
        for on in (1, 0):
            hostcheck.hc_set_origin_skip(on)
            out = np.zeros((h, w, 3), np.float32)
            rays = (C.c_ulonglong * 2)()
            hostcheck.hc_render(w, h, 0, 8, 32, 1, 0, 4, P(out), rays)
            outs.append((out, rays[0], rays[1]))
        hostcheck.hc_set_origin_skip(1)
        assert np.array_equal(outs[0][0], outs[1][0])
        assert outs[0][1:] == outs[1][1:]
    hostcheck.hc_scene_cornell()


# ------------------------------------------------------------------ EXR, trainer stand-in
def test_exr_round_trip(pt, tmp_path):
    rng = np.random.default_rng(0)
    for (h, w) in [(1, 1), (3, 5), (48, 64)]:
        img = rng.uniform(0, 20, (h, w, 3)).astype(np.float32)
        path = str(tmp_path / f"t_{w}x{h}.exr")
        assert pt.write_exr(path, img, 1.0) == 0
        back = pt.read_exr(path)
        assert np.array_equal(back, img)
        assert pt.write_exr(path, img, 0.5) == 0  # src/main.rs:427-434: value * exposure
        assert np.array_equal(pt.read_exr(path), img * np.float32(0.5))
    raw = open(path, "rb").read()
    assert raw[:4] == bytes([0x76, 0x2f, 0x31, 0x01])  # OpenEXR magic
    for attr in (b"channels", b"compression", b"dataWindow", b"displayWindow", b"lineOrder", b"pixelAspectRatio",
                 b"screenWindowCenter", b"screenWindowWidth"):
        assert attr in raw[:512]
    assert pt.lib().pt_write_exr(b"/nonexistent_dir/x.exr", 1, 1, P(np.zeros(3, np.float32)), 1.0) == -1


def test_exr_file_parsed_by_an_independent_reader(pt, tmp_path):
    """The file the writer produces, read by a parser that shares nothing with it (tests/exr_reader.py: the
    OpenEXR single-part scanline layout unpacked field by field): magic and version, the eight required
    attributes with their types and sizes, channels B, G, R as 32-bit floats in alphabetical order, no
    compression, an offset table that points at one chunk per scanline, chunks that fill the file exactly --
    and the pixel values, bit for bit, with the exposure applied (src/main.rs:420-435)."""
    import struct
    import exr_reader
    rng = np.random.default_rng(3)
    for (h, w) in [(1, 1), (2, 7), (54, 96)]:
        img = rng.uniform(0, 20, (h, w, 3)).astype(np.float32)
        img[0, 0] = (0.0, np.float32(1e-30), np.float32(6.5e4))  # values a half-float file would not hold
        path = str(tmp_path / f"i_{w}x{h}.exr")
        for exposure in (1.0, 0.25):
            assert pt.write_exr(path, img, exposure) == 0
            back, attrs = exr_reader.read_rgb(path)
            assert back.shape == (h, w, 3)
            assert np.array_equal(back.view(np.uint32), (img * np.float32(exposure)).view(np.uint32))
        assert struct.unpack("<4i", attrs["dataWindow"][1]) == (0, 0, w - 1, h - 1) == struct.unpack("<4i", attrs["displayWindow"][1])
        assert struct.unpack("<f", attrs["pixelAspectRatio"][1]) == (1.0,) and attrs["lineOrder"][1] == b"\0"
        assert [c[:2] for c in exr_reader.parse_channels(attrs["channels"][1])] == [("B", 2), ("G", 2), ("R", 2)]
        # ... and the two readers agree
        assert np.array_equal(pt.read_exr(path), back)


def _decode_png(raw):
    """A PNG decoder for the files under test: chunks with checked CRCs, one zlib stream, filters 0-2, RGB 8/16."""
    import struct
    import zlib
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(raw):
        n, kind = struct.unpack(">I4s", raw[at:at + 8])
        data = raw[at + 8:at + 8 + n]
        assert struct.unpack(">I", raw[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + data)
        chunks.append((kind, data))
        at += 12 + n
    assert chunks[0][0] == b"IHDR" and chunks[-1] == (b"IEND", b"")
    w, h, bits, colour, comp, flt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (colour, comp, flt, lace) == (2, 0, 0, 0) and bits in (8, 16)
    assert (b"sRGB", b"\x00") in chunks
    data = zlib.decompress(b"".join(d for k, d in chunks if k == b"IDAT"))
    row = w * 3 * bits // 8
    assert len(data) == (row + 1) * h
    out = np.zeros((h, row), np.uint8)
    prev = np.zeros(row, np.uint8)
    for y in range(h):
        line = np.frombuffer(data, np.uint8, row, y * (row + 1) + 1)
        f = data[y * (row + 1)]
        assert f in (0, 2)
        prev = line if f == 0 else (line + prev).astype(np.uint8)
        out[y] = prev
    if bits == 16:
        return (out[:, 0::2].astype(np.uint16) << 8 | out[:, 1::2]).reshape(h, w, 3), bits
    return out.reshape(h, w, 3), bits


def test_png_export_is_the_srgb_encoding_of_the_render(pt, tmp_path):
    """The display copy: value * exposure clamped to [0, 1], sRGB transfer (IEC 61966-2-1) in float64 here, nearest
    code of 8 or 16 bits; negative values and NaN come out black, everything from 1 up white.  The reference's
    own pt.png was written by code no longer in its tree, so the format is checked, not that file's bytes."""
    rng = np.random.default_rng(3)
    for (h, w) in [(1, 1), (5, 7), (54, 96)]:
        img = rng.uniform(0, 1.3, (h, w, 3)).astype(np.float32)
        img.ravel()[:: 11] = rng.uniform(0, 0.01, img.ravel()[:: 11].shape)  # the linear toe
        if img.size > 8:
            img.ravel()[1], img.ravel()[2], img.ravel()[3] = -0.5, np.nan, np.inf
        for bits, exposure in ((8, 1.0), (16, 1.0), (8, 0.5)):
            path = str(tmp_path / f"t_{w}x{h}_{bits}.png")
            assert pt.write_png(path, img, exposure, bits) == 0
            codes, got_bits = _decode_png(open(path, "rb").read())
            assert got_bits == bits and codes.shape == (h, w, 3)
            v = np.nan_to_num(img.astype(np.float64) * exposure, nan=0.0, posinf=1.0).clip(0.0, 1.0)
            enc = np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(v, 1 / 2.4) - 0.055)
            want = enc * (2 ** bits - 1)
            # float32 arithmetic in the writer: the code is the nearest one except within 2e-3 of a code of a tie
            assert np.abs(codes.astype(np.float64) - want).max() < 0.5 + (2e-3 if bits == 8 else 0.05)
            if img.size > 8:
                assert tuple(codes.ravel()[1:4]) == (0, 0, 2 ** bits - 1)
    L = pt.lib()
    assert L.pt_write_png(b"/nonexistent_dir/x.png", 1, 1, P(np.zeros(3, np.float32)), 1.0, 8) == -4
    assert L.pt_write_png(str(tmp_path / "x.png").encode(), 1, 1, P(np.zeros(3, np.float32)), 1.0, 12) == -1
    assert L.pt_write_png(str(tmp_path / "x.png").encode(), 0, 1, P(np.zeros(3, np.float32)), 1.0, 8) == -1
    # the demosaic entry points turn an empty image or a missing buffer down before anything touches the GPU
    z = np.zeros(3, np.float32)
    assert L.pt_demosaic_f32(P(z), 0, 1, 0, 0, P(z)) == -1 and L.pt_demosaic_f32(P(z), 1, -1, 0, 0, P(z)) == -1
    assert L.pt_demosaic_f32(None, 1, 1, 0, 0, P(z)) == -1 and L.pt_demosaic_device(None, 1, 1, 0, 0, None) == -1


def test_exr_reader_rejects_damaged_files(pt, tmp_path):
    """Truncated or corrupted files: an error code, never a read outside the file."""
    L = pt.lib()
    img = np.random.default_rng(1).uniform(0, 4, (6, 7, 3)).astype(np.float32)
    good = str(tmp_path / "good.exr")
    assert pt.write_exr(good, img) == 0
    raw = open(good, "rb").read()
    w, h = C.c_int(), C.c_int()
    out = np.zeros(img.size, np.float32)
    bad = str(tmp_path / "bad.exr")
    rng = np.random.default_rng(2)
    cuts = list(range(0, 40)) + [len(raw) // 3, len(raw) // 2, len(raw) - 1]
    for cut in cuts:  # every prefix that ends inside the header, and a few inside the pixel data
        open(bad, "wb").write(raw[:cut])
        assert L.pt_read_exr(bad.encode(), C.byref(w), C.byref(h), P(out), out.size) != 0
    header_end = raw.index(b"screenWindowWidth") + 40
    for _ in range(300):  # random byte damage in the header: any return code, no crash
        b = bytearray(raw)
        for k in rng.integers(8, header_end, 3):
            b[k] = int(rng.integers(0, 256))
        open(bad, "wb").write(bytes(b))
        L.pt_read_exr(bad.encode(), C.byref(w), C.byref(h), P(out), out.size)
    b = bytearray(raw)  # a scanline offset far outside the file
    table = len(raw) - 6 * (8 + 7 * 12) - 6 * 8
    b[table:table + 8] = (2**64 - 4).to_bytes(8, "little")
    open(bad, "wb").write(bytes(b))
    assert L.pt_read_exr(bad.encode(), C.byref(w), C.byref(h), P(out), out.size) == -2
    assert np.array_equal(pt.read_exr(good), img)


def test_scene_triangles_checks_its_input(pt):
    """pt_scene_triangles takes caller memory: bad counts, null arrays, indices past the vertex array and
    non-finite values are refused (-1) and the scene stays what it was."""
    L = pt.lib()
    L.pt_scene_cornell()
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    I = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    before = pt.scene_arrays()
    u32 = C.POINTER(C.c_uint32)
    assert L.pt_scene_triangles(P(V), -1, I.ctypes.data_as(u32), 2, None) == -1
    assert L.pt_scene_triangles(None, 4, I.ctypes.data_as(u32), 2, None) == -1
    assert L.pt_scene_triangles(P(V), 4, None, 2, None) == -1
    bad = I.copy()
    bad[1, 2] = 4
    assert L.pt_scene_triangles(P(V), 4, bad.ctypes.data_as(u32), 2, None) == -1
    nanv = V.copy()
    nanv[2, 1] = np.nan
    assert L.pt_scene_triangles(P(nanv), 4, I.ctypes.data_as(u32), 2, None) == -1
    M = np.full((2, 11), np.inf, np.float32)
    assert L.pt_scene_triangles(P(V), 4, I.ctypes.data_as(u32), 2, P(M)) == -1
    after = pt.scene_arrays()
    assert all(np.array_equal(a, b) for a, b in zip(before, after))
    assert L.pt_scene_triangles(P(V), 4, I.ctypes.data_as(u32), 2, None) == 0
    assert pt.scene_arrays()[1].shape == (2, 3)
    assert L.pt_scene_triangles(P(V), 0, I.ctypes.data_as(u32), 0, None) == 0  # an empty scene is a scene
    L.pt_scene_cornell()


def test_later_sample_ranges_start_streams_that_differ_between_pixels(hostcheck):
    """The first number a PCG stream yields is a function of the high bits of its start.  The reference's per-pixel
    seed (optix_device.cu:945-948) leaves those the same for every pixel of a frame: kept for the range that starts at
    sample 0, since strict parity needs it.  Every later range (a chunk of a large scene's render can be one sample
    long, a rank's share starts where the previous rank's ends) passes its start through a finaliser: first draws are
    uniform over the pixels and unrelated between neighbours and between consecutive ranges."""
    n = 200_000
    out = np.zeros(n, np.float32)
    hostcheck.hc_first_draws(3, 0, n, P(out))
    assert out.max() - out.min() < 0.01  # the reference's stream start
    draws = []
    for begin in (1, 2, 3, 64, 4095):
        hostcheck.hc_first_draws(3, begin, n, P(out))
        draws.append(out.copy())
        assert abs(out.mean() - 0.5) < 0.005 and abs(out.std() - 12 ** -0.5) < 0.005
        hist = np.histogram(out, bins=64, range=(0, 1))[0]
        assert np.abs(hist - n / 64).max() < 6 * (n / 64) ** 0.5
        assert abs(np.corrcoef(out[:-1], out[1:])[0, 1]) < 0.01  # neighbouring pixels
        assert abs(np.corrcoef(out[:-1920], out[1920:])[0, 1]) < 0.01  # one image row apart
    for a, b in zip(draws[:-1], draws[1:]):
        assert abs(np.corrcoef(a, b)[0, 1]) < 0.01


def test_chunk_plan(pt):
    """Slots per pixel of a fixed-count render.  Scenes that run fused: 32 Mi slots, chunks of 32 samples and more, 8
    at most, the last of them cut into four (of at least 4 samples) so that the render ends on short slots.  Scenes
    that run staged: 144 Mi slots, chunks of one sample and more, 64 at most.  The renderer grows to that if a
    quarter of the free memory (its own 164 bytes per slot counted as free) holds it, else to what does, and never
    shrinks."""
    plan = pt.lib().pt_dbg_chunk_plan
    big = 1 << 40
    # fused scenes
    assert plan(1920, 1080, 1024, 0, 1, big) == 7 + 4   # 7 chunks of 128 samples and 4 of 32
    assert plan(1920, 1080, 100, 0, 1, big) == 2 + 4    # 100 // 3 = 33 >= 32: 2 chunks of 34 samples and 4 of 8
    assert plan(1920, 1080, 63, 0, 1, big) == 1
    assert plan(3840, 2160, 4096, 0, 1, big) == 3 + 4   # 32 Mi / 8.3 M pixels = 4 chunks, the last cut
    assert plan(64, 36, 4096, 0, 1, big) == 7 + 4
    assert plan(1920, 1080, 1024, 0, 8, 0) == 8         # no room for the cut: eight equal chunks
    # staged scenes
    assert plan(1920, 1080, 256, 1, 1, big) == 64
    assert plan(3840, 2160, 16, 1, 1, big) == 16
    assert plan(3840, 2160, 4096, 1, 1, big) == 18      # 144 Mi / 8.3 M pixels
    assert plan(3840, 2160, 2, 1, 1, big) == 2 and plan(3840, 2160, 1, 1, 1, big) == 1
    assert plan(320, 180, 5, 1, 1, big) == 5
    # memory: 1080p, 64 chunks are 21.8 GB
    need = 1920 * 1080 * 64 * 164
    assert plan(1920, 1080, 256, 1, 1, 4 * need) == 64
    assert plan(1920, 1080, 256, 1, 1, 4 * need - 1920 * 1080 * 164 - 1) == 63  # its own chunk counts as free
    assert plan(1920, 1080, 256, 1, 1, 2 * need) == 32
    assert plan(1920, 1080, 256, 1, 4, 0) == 4          # nothing free: what it holds
    assert plan(1920, 1080, 256, 1, 64, 0) == 64
    assert plan(1920, 1080, 8, 1, 64, 0) == 8           # fewer samples than it could hold
    assert plan(0, 1080, 8, 1, 1, big) == -1 and plan(16, 16, 8, 1, 0, big) == -1


def test_shard_ranges_partition_the_samples(pt):
    """pt_shard_range: the in-library shards of a render call are contiguous, disjoint, cover [0, spp) and differ
    by at most one sample; the setting itself needs no GPU."""
    L = pt.lib()
    for spp in (0, 1, 5, 64, 1000, 1024, 4097):
        for n in (1, 2, 3, 4, 8, 16):
            pos, sizes = 0, []
            for d in range(n):
                b, c = C.c_int(), C.c_int()
                L.pt_shard_range(spp, n, d, C.byref(b), C.byref(c))
                assert b.value == pos and c.value >= 0
                pos += c.value
                sizes.append(c.value)
            assert pos == spp and max(sizes) - min(sizes) <= 1
    before = L.pt_get_device_count()
    L.pt_set_device_count(4)
    assert L.pt_get_device_count() == 4
    L.pt_set_device_count(0)
    assert L.pt_get_device_count() == 1
    L.pt_set_device_count(before)


def test_pgl_stand_in_trainer(pt):
    """ABI and ownership of the pgl_* calls (openpgl_bridge.h:7-33); the trainer itself is not OpenPGL."""
    L = pt.lib()
    dev = L.pgl_dev_create(0)
    lo, hi = (C.c_float * 3)(-1, -1, 0), (C.c_float * 3)(1, 1, 2)
    field = L.pgl_field_create(dev, lo, hi)
    samples = L.pgl_samples_create()
    rng = np.random.default_rng(4)
    mu = np.array([0.0, 0.0, 1.0])
    for _ in range(5000):
        pos = rng.uniform([-1, -1, 0], [1, 1, 2])
        d = mu + 0.15 * rng.normal(size=3)
        d /= np.linalg.norm(d)
        L.pgl_samples_add_surface(samples, (C.c_float * 3)(*pos), (C.c_float * 3)(*d), (C.c_float * 3)(0.5, 0.6, 0.7), 0)
    L.pgl_field_update(field, samples)
    regs, lobes = C.POINTER(pt.PglRegion)(), C.POINTER(pt.PglLobe)()
    nr, nl = C.c_uint32(), C.c_uint32()
    L.pgl_field_snapshot(field, C.byref(regs), C.byref(nr), C.byref(lobes), C.byref(nl))
    assert nr.value == 64 and nl.value >= 64
    total = 0
    for i in range(nr.value):
        r = regs[i]
        assert r.lobe_ofs == total
        total += r.lobe_num
        ws = [lobes[r.lobe_ofs + j].weight for j in range(r.lobe_num)]
        assert abs(sum(ws) - 1.0) < 1e-4
        best = max(range(r.lobe_num), key=lambda j: ws[j])
        lb = lobes[r.lobe_ofs + best]
        assert lb.mu[2] > 0.9 and lb.kappa > 5.0
    assert total == nl.value
    L.pgl_samples_destroy(samples)
    L.pgl_field_destroy(field)
    L.pgl_dev_destroy(dev)


# ------------------------------------------------------------------ multi-process reduce (gloo)
_WORKER = r'''
import os, sys, ctypes as C
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from ptamd import sharded
hc = C.CDLL(sys.argv[3])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[4], rank=int(sys.argv[5]), world_size=2)
rank = dist.get_rank()
w, h, spp = 24, 16, 8
def render_chunk(sums, begin, count):
    out = np.zeros(w * h * 3, np.float32)
    hc.hc_render(w, h, begin, count, 32, 1, 0, 1, out.ctypes.data_as(C.POINTER(C.c_float)), None)
    sums += torch.from_numpy(out)
begin, count = sharded.chunk_of(rank, spp)
img = sharded.render_sharded(render_chunk, w, h, begin, count, 2 * spp)
if rank == 0:
    np.save(sys.argv[6], img.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_sample_sharded_reduce_two_ranks(hostcheck, tmp_path):
    """Two gloo ranks each render one chunk (device logic run on the host) and all-reduce the
    sums; the result equals one process rendering both chunks."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    out = str(tmp_path / "img.npy")
    so = os.path.join(HERE, "hostcheck", "libhostcheck.so")
    pkg = os.path.join(ROOT, "rust-pathtracer_amd")
    procs = [subprocess.Popen([sys.executable, str(script), pkg, HERE, so, str(port), str(r), out]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    img = np.load(out)
    w, h, spp = 24, 16, 8
    want = np.zeros((h, w, 3), np.float32)
    for begin in (0, spp):
        o = np.zeros((h, w, 3), np.float32)
        hostcheck.hc_render(w, h, begin, spp, 32, 1, 0, 1, P(o), None)
        want += o
    want /= 2 * spp
    assert np.allclose(img, want, rtol=1e-6, atol=1e-7)
    from ptamd import sharded
    assert sharded.split_samples(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert sharded.chunk_of(3, 512) == (1536, 512)


_SHARED_WORKER = r'''
import sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ptamd import sharded
rank = int(sys.argv[3])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=rank, world_size=2)
w, h = 6, 4
calls = []
def render_share(stats, begin, count, total, reduce):
    # three rounds; in each this rank adds its own statistics on top of the totals of the round before
    local = torch.zeros(w * h, 8)
    for rnd in range(3):
        local[:, 0:3] += (rank + 1) * (rnd + 1)
        local[:, 3] += count / 4
        local[:, 4:7] += ((rank + 1) * (rnd + 1)) ** 2
        stats.copy_(local.reshape(-1))
        reduce()
        calls.append(stats.clone())
begin, count = sharded.chunk_of(rank, 8)
mean, cnt = sharded.render_sharded_adaptive_shared(render_share, w, h, begin, count, 16)
np.savez(sys.argv[4] + "/r%d.npz" % rank, mean=mean.numpy(), cnt=cnt.numpy(), first=calls[0].numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_shared_adaptive_reduce_two_ranks(tmp_path):
    """The reduce step of the shared noise test over two gloo ranks: after every call all ranks hold the sum of
    the ranks' statistics, and the image is sum / samples of the last one."""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    script = tmp_path / "shared_worker.py"
    script.write_text(_SHARED_WORKER)
    pkg = os.path.join(ROOT, "rust-pathtracer_amd")
    procs = [subprocess.Popen([sys.executable, str(script), pkg, str(port), str(r), str(tmp_path)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a, b = (np.load(tmp_path / f"r{r}.npz") for r in range(2))
    for k in ("mean", "cnt", "first"):
        assert np.array_equal(a[k], b[k])
    first = a["first"].reshape(-1, 8)
    assert (first[:, 0] == 3).all() and (first[:, 3] == 4).all() and (first[:, 4] == 5).all()
    # after three rounds: sums (1+2)*(1+2+3) = 18 over 2 * 3 * 2 = 12 samples
    assert (a["cnt"] == 12).all() and np.allclose(a["mean"], 1.5)


# ------------------------------------------------------------------ scene input (OBJ / MTL)
_OBJ = """# a quad with texture/normal references, a pentagon with negative indices, a glass triangle
mtllib things.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0 0
vn 0 0 1
usemtl red
f 1/1/1 2/1/1 3/1/1 4/1/1
v 2 0 1
v 3 0 1
v 3.5 1 1
v 2.5 2 1
v 1.5 1 1
usemtl lamp
f -5 -4 -3 -2 -1
usemtl glass
f 1//1 5//1 9//1
usemtl missing
f 1 2 5
"""
_MTL = """newmtl red
Kd 0.5 0.1 0.05
Ns 98
newmtl lamp
Kd 0 0 0
Ke 7 8 9
Pr 0.6
Ns 10
newmtl glass
Kd 1 1 1
Ni 1.33
d 0.25
Pr 0
"""


def test_obj_scene_input(pt, tmp_path):
    """Wavefront OBJ in: polygons are fanned from their first corner into consecutive triangles, index forms
    and negative indices resolve, materials come from the library (missing ones fall back to the default)."""
    (tmp_path / "things.obj").write_text(_OBJ)
    (tmp_path / "things.mtl").write_text(_MTL)
    L = pt.lib()
    try:
        assert L.pt_scene_load_obj(str(tmp_path / "things.obj").encode()) == 2 + 3 + 1 + 1
        V, I, N, M = pt.scene_arrays()
        assert V.shape == (9, 3) and I.tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [4, 7, 8], [0, 4, 8], [0, 1, 4]]
        assert np.allclose(N[0], [0, 0, 1]) and np.allclose(N[1], [0, 0, 1])
        red, lamp, glass, default = M[0], M[2], M[5], M[6]
        assert np.array_equal(M[0], M[1]) and np.array_equal(M[2], M[4])
        assert np.allclose(red[0:3], [0.5, 0.1, 0.05]) and red[4] == pytest.approx(np.sqrt(2 / 100.0)) and red[7] == 0
        assert np.allclose(lamp[8:11], [7, 8, 9]) and lamp[4] == pytest.approx(0.6)  # Pr wins over Ns
        assert glass[5] == pytest.approx(1.33) and glass[7] == pytest.approx(0.75) and glass[4] == 0
        assert np.allclose(default[0:3], [0.789, 0.801, 0.8]) and default[4] == pytest.approx(0.25) and default[5] == pytest.approx(1.5)
        # errors leave the scene as it was
        assert L.pt_scene_load_obj(str(tmp_path / "nothing.obj").encode()) == -1
        (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nf 1 2 7\n")
        assert L.pt_scene_load_obj(str(tmp_path / "bad.obj").encode()) == -1
        assert pt.scene_arrays()[1].shape == (7, 3)
    finally:
        L.pt_scene_cornell()


def test_obj_round_trip_is_exact(pt, tmp_path):
    """The built-in scene written as OBJ + MTL and read back is the same scene, bit for bit."""
    L = pt.lib()
    L.pt_scene_cornell()
    want = pt.scene_arrays()
    path = str(tmp_path / "cornell.obj").encode()
    assert L.pt_scene_save_obj(path) == 0
    try:
        assert L.pt_scene_load_obj(path) == 24
        got = pt.scene_arrays()
        # the file holds only the vertices, so duplicates stay duplicates and the order is kept
        for a, b in zip(want, got):
            assert a.shape == b.shape and np.array_equal(a, b)
    finally:
        L.pt_scene_cornell()


# ------------------------------------------------------------------ guiding across ranks (gloo)
_GUIDE_WORKER = r'''
import sys
import numpy as np, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import ptamd
from ptamd import sharded
rank = int(sys.argv[3])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=rank, world_size=2)
rng = np.random.default_rng(rank)
n = 3000 + 1000 * rank  # rings of different lengths
rec = np.zeros((n, 10), np.float32)
rec[:, 0:2] = rng.uniform(-1, 1, (n, 2)); rec[:, 2] = rng.uniform(0, 2, n)
d = rng.normal(size=(n, 3)) * 0.05
d[:, 2] += 1.0 if rank == 0 else -1.0  # rank 0 sees light from above, rank 1 from below
rec[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
rec[:, 6:9] = 1.0
got = {}
g = sharded.ShardedGuiding(ptamd, (-1, -1, 0), (1, 1, 2), collect=lambda: rec,
                           upload=lambda r, l: got.update(regions=r.copy(), lobes=l.copy()), reset=lambda: got.update(reset=True))
regions, lobes = g.process_batch()
g.close()
assert got["reset"] and np.array_equal(got["regions"], regions)
np.savez(sys.argv[4] + "/g%d.npz" % rank, regions=regions, lobes=lobes)
dist.barrier()
dist.destroy_process_group()
'''


def test_guiding_across_two_ranks(pt, tmp_path):
    """src/guiding.rs:222-247 with two ranks: both rings reach the one field on rank 0, and both ranks end up
    with the same snapshot, which holds what either rank saw (one lobe up, one lobe down)."""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    script = tmp_path / "guide_worker.py"
    script.write_text(_GUIDE_WORKER)
    pkg = os.path.join(ROOT, "rust-pathtracer_amd")
    procs = [subprocess.Popen([sys.executable, str(script), pkg, str(port), str(r), str(tmp_path)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a, b = (np.load(tmp_path / f"g{r}.npz") for r in range(2))
    assert np.array_equal(a["regions"], b["regions"]) and np.array_equal(a["lobes"], b["lobes"])
    regions, lobes = a["regions"], a["lobes"]
    assert len(regions) >= 1 and len(lobes) >= 2
    assert regions[:, 6].astype(int).tolist() == np.concatenate([[0], np.cumsum(regions[:-1, 7])]).astype(int).tolist()
    up = lobes[(lobes[:, 2] > 0.9) & (lobes[:, 4] > 0.2)]
    down = lobes[(lobes[:, 2] < -0.9) & (lobes[:, 4] > 0.2)]
    assert len(up) >= 1 and len(down) >= 1  # the field was trained on the union of the two rings
    # one rank alone (no process group): the same calls, no collective
    from ptamd import sharded
    rec = np.zeros((500, 10), np.float32)
    rec[:, 5] = 1.0
    rec[:, 6:9] = 1.0
    seen = {}
    g = sharded.ShardedGuiding(pt, (-1, -1, 0), (1, 1, 2), collect=lambda: rec, upload=lambda r, l: seen.update(l=l),
                               reset=lambda: None)
    r1, l1 = g.process_batch()
    g.close()
    assert len(l1) >= 1 and seen["l"] is l1 and l1[np.argmax(l1[:, 4]), 2] > 0.9


def test_batched_lobe_passes_are_the_loops(hostcheck):
    """Regions of up to four lobes are read in batches on the shading path; the lobe chosen and the mixture pdf
    are those of the lobe-by-lobe loops, bit for bit -- also with zero weights, with one lobe, and with running
    sums that repeat (the first lobe that reaches u x total wins in both)."""
    rng = np.random.default_rng(17)
    for trial in range(3000):
        n = int(rng.integers(1, 5))
        lobes = np.zeros((n, 8), np.float32)
        mu = rng.normal(size=(n, 3))
        lobes[:, 0:3] = mu / np.linalg.norm(mu, axis=1, keepdims=True)
        lobes[:, 3] = rng.choice([0.0, 1e-5, 0.5, 7.0, 60.0], n)
        lobes[:, 4] = rng.choice([0.0, 0.0, 1.0, 0.25], n) * rng.uniform(0.5, 1.0, n)
        lobes[:, 5:8] = 1.0
        wi = rng.normal(size=3)
        wi = (wi / np.linalg.norm(wi)).astype(np.float32)
        u = float(rng.choice([0.0, 0.25, 0.5, 0.999999, float(rng.uniform())]))
        il, ib = C.c_int(), C.c_int()
        pl, pb = C.c_float(), C.c_float()
        hostcheck.hc_guide_lobe_passes(P(lobes), n, C.c_float(u), P(wi), C.byref(il), C.byref(ib), C.byref(pl), C.byref(pb))
        assert il.value == ib.value, (trial, lobes, u)
        assert np.float32(pl.value).tobytes() == np.float32(pb.value).tobytes(), (trial, pl.value, pb.value)


# ------------------------------------------------------------------ training statistics binned per region and lobe
def _random_records(rng, n):
    rec = np.zeros((n, 10), np.float32)
    rec[:, 0:2] = rng.uniform(-1.1, 1.1, (n, 2))  # some fall outside the field
    rec[:, 2] = rng.uniform(-0.1, 2.1, n)
    d = rng.normal(size=(n, 3))
    rec[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rec[:, 6:9] = rng.uniform(0, 2, (n, 3)) * (rng.uniform(size=(n, 1)) > 0.05)  # a few with weight zero
    return rec


def test_binned_update_is_the_sample_update(pt):
    """pgl_field_update fed record by record and pt_pgl_field_update_binned fed the per-(region, lobe) fixed-point
    sums of the same records (tests/binning.py restates the device kernel) train the same field: twice in a row,
    so the second batch is assigned by the axes the first one moved."""
    import binning
    rng = np.random.default_rng(5)
    a, b = pt.GuidingField((-1, -1, 0), (1, 1, 2)), pt.GuidingField((-1, -1, 0), (1, 1, 2))
    try:
        for _ in range(2):
            rec = _random_records(rng, 4000)
            bmin, bmax, res, axes = b.layout()
            a.update(rec)
            b.update_binned(binning.bin_records(rec, bmin, bmax, res, axes))
            (ra, la), (rb, lb) = a.snapshot(), b.snapshot()
            assert np.array_equal(ra, rb) and la.shape == lb.shape
            assert np.allclose(la, lb, rtol=2e-5, atol=2e-6), np.abs(la - lb).max()
            assert np.allclose(a.layout()[3], b.layout()[3], atol=1e-6)  # the assignment axes (2^-24 fixed point against doubles)
    finally:
        a.close()
        b.close()


def test_device_binning_step_is_the_numpy_restatement(hostcheck, pt):
    """bin_train_record, the per-record body of the GPU kernel, compiled for the host: over random records (some
    outside the field, some of weight zero, some not numbers, some with a digest that does not match) the table is the
    one tests/binning.py builds, word for word."""
    import binning
    rng = np.random.default_rng(9)
    f = pt.GuidingField((-1, -1, 0), (1, 1, 2))
    try:
        f.update(_random_records(rng, 3000))  # move the axes off the coordinate directions
        bmin, bmax, res, axes = f.layout()
        rec = _random_records(rng, 5000)
        rec[7, 0], rec[8, 2], rec[9, 7] = np.nan, np.inf, np.nan
        torn = (rng.uniform(size=len(rec)) < 0.02).astype(np.int32)
        table = np.zeros(int(np.prod(res)) * 6 * 8, np.int64)
        hostcheck.hc_bin_records(P(rec), len(rec), P(bmin), P(bmax), P(res, C.c_int), P(axes), P(torn, C.c_int),
                                 table.ctypes.data_as(C.POINTER(C.c_longlong)))
        want = binning.bin_records(rec[torn == 0], bmin, bmax, res, axes)
        assert np.array_equal(table, want) and table[7::8].sum() > 3000
    finally:
        f.close()


def test_trainer_drops_records_that_are_not_numbers(pt):
    """A position or weight that is not a finite number trains nothing, on either path (the comparison against the
    field's bounds must not let a NaN through to the cell index)."""
    import binning
    f = pt.GuidingField((-1, -1, 0), (1, 1, 2))
    try:
        rec = np.zeros((4, 10), np.float32)
        rec[:, 5] = 1.0
        rec[:, 6:9] = 1.0
        rec[0, 0], rec[1, 1], rec[2, 6] = np.nan, np.inf, np.nan
        assert binning.bin_records(rec, *f.layout())[7::8].sum() == 1
        f.update(rec)
        regions, lobes = f.snapshot()
        assert len(regions) == 64 and len(lobes) == 1 and lobes[0, 2] > 0.99
    finally:
        f.close()


def test_trainer_drops_weights_the_table_cannot_hold(pt, hostcheck):
    """A firefly of 1e12 would overflow the 64-bit fixed-point sums of the binned form (2^-24 steps) and a weight
    below half a step would count as a sample of weight zero there, while the record-by-record form works in
    doubles: both forms drop records outside [PT_PGL_MIN_WEIGHT, PT_PGL_MAX_WEIGHT] (pt_api.h), so the fields they
    train stay the same field.  Checked on the numpy restatement, on the kernel's per-record body compiled for the
    host, and on both update calls of the library."""
    import binning
    rec = np.zeros((6, 10), np.float32)
    rec[:, 0:3] = (0.1, 0.2, 0.3)
    rec[:, 5] = 1.0           # dir_in = +z
    rec[:, 6:9] = 0.5         # an ordinary record ...
    rec[1, 6:9] = 1e12        # ... a firefly
    rec[2, 6:9] = (3e6, 0.0, 0.0)   # mean 1e6 is inside the range, its red component is not
    rec[3, 6:9] = 1e-9        # rounds to no step of the table
    rec[4, 6:9] = 2.0 ** 20   # the largest weight that is used
    rec[5, 6:9] = (-4.0, 1.0, 4.5)  # negative components are fine while the mean is positive
    used = [0, 4, 5]
    a, b = pt.GuidingField((-1, -1, 0), (1, 1, 2)), pt.GuidingField((-1, -1, 0), (1, 1, 2))
    try:
        lay = a.layout()
        table = binning.bin_records(rec, *lay)
        assert table[7::8].sum() == len(used)
        assert table[3::8].sum() == sum(int(np.rint(float(rec[i, 6:9].astype(np.float32).sum() / np.float32(3)) * 2.0 ** 24)) for i in used)
        # the device kernel's body, record by record
        bmin, bmax, res, axes = lay
        got = np.zeros_like(table)
        torn = np.zeros(len(rec), np.int32)
        hostcheck.hc_bin_records(P(rec), len(rec), P(bmin), P(bmax), P(res, C.c_int), P(axes), P(torn, C.c_int),
                                 got.ctypes.data_as(C.POINTER(C.c_longlong)))
        assert np.array_equal(got, table)
        # both update calls train the same field from these records
        a.update(rec)
        b.update_binned(table)
        (ra, la), (rb, lb) = a.snapshot(), b.snapshot()
        assert np.array_equal(ra, rb) and la.shape == lb.shape and np.allclose(la, lb, rtol=2e-5, atol=2e-6)
    finally:
        a.close()
        b.close()


_BINNED_WORKER = r'''
import sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[5])
import ptamd, binning
from ptamd import sharded
rank = int(sys.argv[3])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=rank, world_size=2)
rng = np.random.default_rng(100 + rank)
out = []
g = sharded.BinnedGuiding(ptamd, (-1, -1, 0), (1, 1, 2), upload=lambda r, l: None, reset=lambda: None,
                          bin=lambda field: torch.from_numpy(binning.bin_records(recs, *field.layout())))
for batch in range(2):
    n = 2000 + 700 * rank
    recs = np.zeros((n, 10), np.float32)
    recs[:, 0:2] = rng.uniform(-1, 1, (n, 2)); recs[:, 2] = rng.uniform(0, 2, n)
    d = rng.normal(size=(n, 3)); recs[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    recs[:, 6:9] = rng.uniform(0, 2, (n, 3))
    regions, lobes = g.process_batch()
    out.append((recs, regions, lobes))
g.close()
np.savez(sys.argv[4] + "/b%d.npz" % rank, **{"%s%d" % (k, i): v for i, t in enumerate(out) for k, v in zip(("rec", "regions", "lobes"), t)})
dist.barrier()
dist.destroy_process_group()
'''


def test_binned_guiding_across_two_ranks(pt, tmp_path):
    """Every rank keeps a copy of the field and the ranks add their int64 tables with one all-reduce: after two
    batches both copies are the same bits, and they are the field one process trains on the union of the records."""
    import socket
    import binning
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    script = tmp_path / "binned_worker.py"
    script.write_text(_BINNED_WORKER)
    pkg = os.path.join(ROOT, "rust-pathtracer_amd")
    procs = [subprocess.Popen([sys.executable, str(script), pkg, str(port), str(r), str(tmp_path), os.path.join(ROOT, "tests")])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a, b = (np.load(tmp_path / f"b{r}.npz") for r in range(2))
    one = pt.GuidingField((-1, -1, 0), (1, 1, 2))
    try:
        for i in range(2):
            assert np.array_equal(a[f"regions{i}"], b[f"regions{i}"]) and np.array_equal(a[f"lobes{i}"], b[f"lobes{i}"])
            union = np.concatenate([a[f"rec{i}"], b[f"rec{i}"]])
            one.update_binned(binning.bin_records(union, *one.layout()))
            regions, lobes = one.snapshot()
            assert np.array_equal(regions, a[f"regions{i}"]) and np.array_equal(lobes, a[f"lobes{i}"])
    finally:
        one.close()


# ------------------------------------------------------------------ scenes made of instances (two-level tree)
def _hc_trace_instanced(hc, o, d, tmax=None, any_hit=0):
    n = len(o)
    tmin = np.zeros(n, np.float32)
    tmax = np.full(n, 1e16, np.float32) if tmax is None else tmax
    prim, inst = np.zeros(n, np.int32), np.zeros(n, np.int32)
    t, b1, b2 = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    hc.hc_trace_rays_instanced(n, P(o), P(d), P(tmin), P(tmax), any_hit, P(prim, C.c_int), P(t), P(b1), P(b2), P(inst, C.c_int))
    return prim, t, b1, b2, inst


def test_instanced_scene_flattens_to_the_flat_one(hostcheck):
    """The synthetic scene as instances (room placed once, one unit icosphere placed per object with a scale, a
    translation and a colour) written out in world space is the flat synthetic scene: same triangles, same materials."""
    hostcheck.hc_scene_synthetic(7, 27, 2)
    try:
        want = _scene_arrays(hostcheck)
        hostcheck.hc_scene_synthetic_instanced(7, 27, 2)
        assert hostcheck.hc_num_instances() == 28
        nt = C.c_int(); nn = C.c_int(); dp = C.c_int()
        hostcheck.hc_scene_info(C.byref(nt), C.byref(nn), C.byref(dp))
        assert nt.value == 12 + 320  # the meshes: one room, one sphere
        hostcheck.hc_scene_flatten()
        got = _scene_arrays(hostcheck)
        V0, I0, N0, M0 = want[:4]
        V1, I1, N1, M1 = got[:4]
        assert len(I0) == len(I1) and np.array_equal(V0[I0], V1[I1]) and np.array_equal(M0, M1)
    finally:
        hostcheck.hc_scene_cornell()


def test_instanced_traversal_matches_brute_force(hostcheck, oracle):
    """Rays through the two-level tree (top level over instance boxes, the ray taken to object space inside an
    instance) find what the oracle's brute-force loop finds in the same scene written out as world triangles:
    the same triangle of the same instance at the same distance, and the same occlusion."""
    hostcheck.hc_scene_synthetic_instanced(7, 64, 2)
    try:
        o, d = _random_rays(3000, 21)
        prim, t, b1, b2, inst = _hc_trace_instanced(hostcheck, o, d)
        tmax = np.random.default_rng(22).uniform(0.05, 2.0, len(o)).astype(np.float32)
        occ = _hc_trace_instanced(hostcheck, o, d, tmax, any_hit=1)[0]
        hostcheck.hc_scene_flatten()
        V, I, N, M, nodes, depth = _scene_arrays(hostcheck)
        assert len(I) == 12 + 64 * 320
        oracle.lib().orc_scene_set(P(V), len(V), P(I, C.c_uint32), len(I), None)
        op, ot = _orc_trace(oracle, o, d)
        # flat triangle index of (instance, triangle of the meshes): instance 0 is the room, the others one sphere each
        flat = np.where(inst == 0, prim, 12 + (inst - 1) * 320 + (prim - 12))
        flat = np.where(prim < 0, -1, flat)
        assert (flat != op).sum() == 0
        assert np.abs(t - ot)[op >= 0].max() < 2e-5
        assert ((b1 >= -1e-5) & (b2 >= -1e-5) & (b1 + b2 <= 1.0 + 1e-5))[op >= 0].all()
        L = oracle.lib()
        fp = C.POINTER(C.c_float)
        want = np.array([L.orc_trace_any(o[i].ctypes.data_as(fp), d[i].ctypes.data_as(fp), 0.0, float(tmax[i]))
                         for i in range(len(o))])
        assert (occ != want).sum() <= 2  # a blocker within rounding of tmax may fall either side in object space
        # the flat tree over the same triangles agrees too
        fprim, ft, _, _ = _hc_trace(hostcheck, o, d)
        assert np.array_equal(fprim, op) and np.abs(ft - t)[op >= 0].max() < 2e-5
    finally:
        oracle.lib().orc_scene_set(None, 0, None, 0, None)
        hostcheck.hc_scene_cornell()


def test_instances_under_rotation_and_uneven_scale(hostcheck, oracle):
    """General affine placements: rotated, unevenly scaled and mirrored copies of one mesh, overlapping each other."""
    rng = np.random.default_rng(31)
    # one mesh: a unit octahedron
    V = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    I = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.uint32)
    xf = []
    for k in range(40):
        a = rng.normal(size=(3, 3))
        q, _ = np.linalg.qr(a)
        m = q @ np.diag(rng.uniform(0.05, 0.4, 3) * rng.choice([-1, 1], 3))
        tr = rng.uniform(-0.8, 0.8, 3) + [0, 0, 1]
        xf.append(np.concatenate([m, tr[:, None]], 1).reshape(-1))
    xf = np.array(xf, np.float32)
    assert hostcheck.hc_scene_instances(P(V), len(V), P(I, C.c_uint32), len(I), len(xf), P(xf)) == 40
    try:
        o, d = _random_rays(4000, 32)
        prim, t, b1, b2, inst = _hc_trace_instanced(hostcheck, o, d)
        hostcheck.hc_scene_flatten()
        FV, FI, _, _, _, _ = _scene_arrays(hostcheck)
        oracle.lib().orc_scene_set(P(FV), len(FV), P(FI, C.c_uint32), len(FI), None)
        op, ot = _orc_trace(oracle, o, d)
        flat = np.where(prim < 0, -1, inst * 8 + prim)
        bad = flat != op
        # two overlapping copies can be hit at distances that differ by rounding only: then either is the closest
        assert (np.abs(t - ot)[bad & (op >= 0) & (prim >= 0)] < 1e-5).all() and ((op >= 0) == (prim >= 0)).all()
        assert bad.mean() < 0.002 and (op >= 0).mean() > 0.3
        assert np.abs(t - ot)[op >= 0].max() < 3e-5
        # the point the hit describes (barycentrics on the world-space triangle) lies on the ray
        hit = (op >= 0) & ~bad
        tri = FV[FI[flat[hit]]]
        pt_ = tri[:, 0] * (1 - b1[hit] - b2[hit])[:, None] + tri[:, 1] * b1[hit][:, None] + tri[:, 2] * b2[hit][:, None]
        assert np.abs(pt_ - (o[hit] + d[hit] * t[hit][:, None])).max() < 1e-4
        # a transform that cannot be inverted is refused
        bad_xf = xf[:1].copy()
        bad_xf[0, 0:3] = 0
        assert hostcheck.hc_scene_instances(P(V), len(V), P(I, C.c_uint32), len(I), 1, P(bad_xf)) == -1
    finally:
        oracle.lib().orc_scene_set(None, 0, None, 0, None)
        hostcheck.hc_scene_cornell()


def test_instanced_closed_mesh_does_not_leak(hostcheck):
    """Rays aimed from inside a closed mesh at its vertices and along its edges never get out, whatever the
    instance's scale: the ray keeps its parameter in object space, so its direction there is not a unit vector
    and the fast test's error band has to grow with its length for points near an edge to reach the watertight test."""
    rng = np.random.default_rng(41)
    V = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    I = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.uint32)
    edges = sorted({tuple(sorted((int(t[a]), int(t[b])))) for t in I for a, b in ((0, 1), (1, 2), (2, 0))})
    try:
        for scale in (1e-3, 0.03, 1.0, 40.0, 3e3):
            a = rng.normal(size=(3, 3))
            q, _ = np.linalg.qr(a)
            m = (q * scale).astype(np.float32)
            centre = rng.uniform(-0.5, 0.5, 3).astype(np.float32)
            xf = np.concatenate([m, centre[:, None]], 1).reshape(1, 12).astype(np.float32)
            assert hostcheck.hc_scene_instances(P(V), len(V), P(I, C.c_uint32), len(I), 1, P(xf)) == 1
            W = V @ m.T + centre  # world-space vertices
            targets = [W[i] for i in range(6)]
            for i, j in edges:
                for s in np.linspace(0.02, 0.98, 33):
                    targets.append(W[i] * (1 - s) + W[j] * s)
            targets = np.array(targets, np.float32)
            o, d = [], []
            for tgt in targets:
                for _ in range(6):
                    dirn = rng.normal(size=3)
                    dirn /= np.linalg.norm(dirn)
                    org = centre + dirn * scale * rng.uniform(0.0, 0.25)  # well inside the unit octahedron's image
                    dd = tgt - org
                    o.append(org)
                    d.append(dd / np.linalg.norm(dd))
            o, d = np.array(o, np.float32), np.array(d, np.float32)
            prim, t, b1, b2, inst = _hc_trace_instanced(hostcheck, o, d)
            assert (prim >= 0).all(), (scale, int((prim < 0).sum()), len(prim))
            # and so does any other ray from the centre
            dirs = rng.normal(size=(2000, 3)).astype(np.float32)
            dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
            prim, _, _, _, _ = _hc_trace_instanced(hostcheck, np.tile(centre, (len(dirs), 1)), dirs)
            assert (prim >= 0).all(), scale
    finally:
        hostcheck.hc_scene_cornell()



# ------------------------------------------------------------------ demosaic: the interior-tile routine
def test_demosaic_interior_tiles_are_the_oracles_bits(hostcheck, oracle):
    """The gradient demosaic's routine for tiles inside the image (pt_demosaic_tile.h: colours fixed by the pattern,
    no clamping, the six passes of src/main.rs:190-342 folded into three) against the oracle's pass-by-pass loops,
    bit for bit, for every pattern; and the tile grid the kernel launches (border tiles listed first, every tile
    exactly once)."""
    rng = np.random.default_rng(21)
    hostcheck.hc_demosaic_interior.restype = C.c_long
    for (h, w) in [(135, 241), (200, 333), (64, 124), (59, 300), (300, 123)]:
        five = np.zeros(5, np.int32)
        hostcheck.hc_demosaic_tile_grid(w, h, P(five, C.c_int))
        gw, gh, ix1, iy1, n_edge = (int(v) for v in five)
        assert gw == -(-w // 60) and gh == -(-h // 28)
        interior = [(bx, by) for by in range(1, iy1) for bx in range(1, ix1)]
        assert n_edge == gw * gh - len(interior)
        for bx, by in interior:  # the apron of four lies inside the image
            assert bx * 60 - 4 >= 0 and by * 28 - 4 >= 0 and bx * 60 + 64 <= w and by * 28 + 32 <= h
        # no tile that could be interior is left to the border routine for nothing
        for by in range(gh):
            for bx in range(gw):
                fits = bx >= 1 and by >= 1 and bx * 60 + 64 <= w and by * 28 + 32 <= h
                assert fits == ((bx, by) in interior) or not interior
        for pattern in range(4):
            raw = rng.uniform(0, 4, (h, w)).astype(np.float32)
            raw[rng.uniform(size=(h, w)) < 0.05] = 0.0
            out = np.full((h, w, 3), -1.0, np.float32)
            covered = np.zeros((h, w), np.uint8)
            n = hostcheck.hc_demosaic_interior(P(raw), w, h, pattern, P(out), P(covered, C.c_ubyte))
            assert n == len(interior) * 60 * 28 == int(covered.sum())
            want = oracle.demosaic(raw, pattern, "amaze")
            m = covered.astype(bool)
            assert np.array_equal(out[m], want[m]), (h, w, pattern)
            assert (out[~m] == -1.0).all()
    # a flat field with zeros sprinkled in (gradients of exactly zero, weights of 1e6) keeps its bits too
    raw = np.full((150, 260), 0.75, np.float32)
    raw[::7, ::5] = 0.0
    for pattern in range(4):
        out = np.zeros((150, 260, 3), np.float32)
        covered = np.zeros((150, 260), np.uint8)
        assert hostcheck.hc_demosaic_interior(P(raw), 260, 150, pattern, P(out), P(covered, C.c_ubyte)) > 0
        m = covered.astype(bool)
        assert np.array_equal(out[m], oracle.demosaic(raw, pattern, "amaze")[m])
