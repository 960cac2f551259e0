"""GPU parity tests: the HIP path, called through the C ABI of libpt_hip.so, against the CPU
oracle and the committed golden vectors.  Bit-exact for integer work (RNG, indices, demosaic),
stated tolerances for floating point."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from test_host_logic import HcRec, P, check_shade_against_oracle, make_shade_cases, _orc_trace, _random_rays

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)


DEFER_DEFAULT = -1  # per scene: shadow rays in place for the built-in scene (pt_set_defer_shadow; 0 in place, 1 queued)

def rel_rmse(img, ref):
    """sqrt(mean(((I-R)/(R+0.01))^2)) per the measurement plan."""
    return float(np.sqrt(np.mean(((img - ref) / (ref + 0.01)) ** 2)))


def tent_downsample(img, f):
    """Image reduced f times per axis with the triangle filter that the renderer's pixel jitter is (sample_tri,
    optix_device.cu:187-196: a tent of one pixel to either side): what a render f times smaller of the same view
    converges to, which a plain block mean is not (it differs along every edge in the picture)."""
    def weights(n_hi):
        n_lo = n_hi // f
        x = (np.arange(n_hi) + 0.5) / f
        w = np.maximum(0.0, 1.0 - np.abs(x[None, :] - (np.arange(n_lo) + 0.5)[:, None]))
        return (w / f).astype(np.float32)
    a = np.tensordot(weights(img.shape[0]), img, axes=(1, 0))           # rows
    return np.moveaxis(np.tensordot(weights(img.shape[1]), a, axes=(1, 1)), 0, 1)  # columns


def rel_rmse_trimmed(img, ref, drop=0.002):
    """rel_rmse without the largest `drop` of the squared relative errors: for estimators with a heavy tail (the
    guided branch), where one firefly in either image is most of the plain figure (0.37 against 0.15 without)."""
    e = np.sort((((img - ref) / (ref + 0.01)) ** 2).ravel())
    return float(np.sqrt(e[: len(e) - int(len(e) * drop)].mean()))


# ------------------------------------------------------------------ a3 / a4 / a5 / a12
def test_rng_bit_exact(gpu, oracle):
    L, OL = gpu.lib(), oracle.lib()
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    for seed, rec in kat["pcg"].items():
        u = np.zeros(16, np.uint32)
        f = np.zeros(16, np.float32)
        L.pt_dbg_pcg(int(seed), 16, P(u, C.c_uint32), P(f))
        assert u.tolist() == rec["u32"]
        assert f.tolist() == [float(np.float32(x)) for x in rec["f32"]]
    s = C.c_uint64(0xdeadbeefcafef00d)
    n = 4096
    u = np.zeros(n, np.uint32)
    f = np.zeros(n, np.float32)
    L.pt_dbg_pcg(s.value, n, P(u, C.c_uint32), P(f))
    want = np.array([OL.orc_pcg(C.byref(s)) for _ in range(n)], np.uint32)
    assert np.array_equal(u, want)


def test_blue_noise_and_jitter(gpu, oracle):
    L, OL = gpu.lib(), oracle.lib()
    rng = np.random.default_rng(0)
    n = 4096
    q = np.stack([rng.integers(0, 3840, n), rng.integers(0, 2160, n), rng.integers(0, 4096, n), rng.integers(1, 9, n)], 1).astype(np.int32)
    uv = np.zeros((n, 2), np.float32)
    jit = np.zeros((n, 2), np.float32)
    L.pt_dbg_blue_noise(n, P(q, C.c_int), P(uv), P(jit))
    wuv = np.zeros((n, 2), np.float32)
    wjit = np.zeros((n, 2), np.float32)
    for i in range(n):
        a, b = (C.c_float * 2)(), (C.c_float * 2)()
        OL.orc_blue_noise(int(q[i, 0]), int(q[i, 1]), int(q[i, 2]), int(q[i, 3]), a)
        OL.orc_olpf_jitter(int(q[i, 0]), int(q[i, 1]), int(q[i, 2]), int(q[i, 3]), b)
        wuv[i], wjit[i] = list(a), list(b)
    # What the device computes, exactly: the hash is integer arithmetic; u = fract(rx + (s + 0.5) * c) with the
    # product and the sum fused into one FMA (one rounding instead of the oracle's two).  Modelled in float64
    # (the sum of a 16-bit and a 48-bit number is exact there, so rounding it to float32 IS the FMA).
    def hash_u32(x):
        x = np.uint64(x)
        m = np.uint64(0xffffffff)
        for shift, mul in ((17, 0xed5ad4bb), (11, 0xac4c1b51), (15, 0x31848bab)):
            x ^= x >> np.uint64(shift)
            x = (x * np.uint64(mul)) & m
        return x ^ (x >> np.uint64(14))
    hv = np.array([int(hash_u32((int(x) * 1973 + int(y) * 9277 + int(f) * 26699) & 0xffffffff)) for x, y, _, f in q])
    r = np.stack([(hv & 0xffff) / 65536.0, (hv >> 16) / 65536.0], 1)
    c32 = np.array([np.float32(1.0) / np.float32(1.32471795724474602596),
                    np.float32(1.0) / (np.float32(1.32471795724474602596) * np.float32(1.32471795724474602596))], np.float32)
    fs = (q[:, 2].astype(np.float32) + np.float32(0.5)).astype(np.float64)
    fused = (r + fs[:, None] * c32.astype(np.float64)[None, :]).astype(np.float32)
    model = fused - np.floor(fused)
    assert np.array_equal(uv, model)  # bit for bit, all 8192 values
    # against the oracle (product rounded, then sum rounded): equal except where that extra rounding shows;
    # measured on the circle, since fract() of a value next to an integer may land on either side
    duv = np.abs(uv - wuv)
    duv = np.minimum(duv, 1.0 - duv)
    assert duv.max() < 5e-4  # one rounding of a value below 4096
    exact = np.mean(uv == wuv)
    assert exact > 0.7, exact  # measured: 0.78 of the values are the oracle's bits
    # jitter: sample_tri of the device's own u with a 1-ulp hardware square root
    u64 = uv.astype(np.float64)
    tri = np.where(u64 < 0.5, np.sqrt(2.0 * u64) - 1.0, 1.0 - np.sqrt(2.0 * (1.0 - u64)))
    assert np.abs(jit - tri).max() < 2.5e-7
    same_side = (np.abs(uv - wuv) < 0.5).all(axis=1)
    crossing = (np.abs(uv - 0.5) < 1e-3).any(axis=1)  # sample_tri has a square-root kink at 0.5
    sel = same_side & ~crossing
    assert np.abs(jit - wjit)[sel].max() < 2e-2 and np.median(np.abs(jit - wjit)) < 1e-6
    assert (uv >= 0).all() and (uv < 1).all() and (np.abs(jit) <= 1).all()


def test_camera_dirs(gpu):
    L = gpu.lib()
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    for cam in kat["camera"]:
        xy = np.array([d["xy"] for d in cam["dirs"]], np.int32)
        jit = np.array([d["jitter"] for d in cam["dirs"]], np.float32)
        out = np.zeros((len(xy), 3), np.float32)
        L.pt_dbg_camera_dirs(cam["w"], cam["h"], len(xy), P(xy, C.c_int), P(jit), P(out))
        want = np.array([d["dir"] for d in cam["dirs"]], np.float32)
        assert np.abs(out - want).max() < 2e-6
        assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-6


def test_scene_upload_is_reference_scene(gpu, oracle):
    L, OL = gpu.lib(), oracle.lib()
    info = gpu.scene_info()
    assert info["triangles"] == 24 and info["nodes"] <= 8 and info["depth"] <= 3
    V = np.zeros((48, 3), np.float32)
    I = np.zeros((24, 3), np.uint32)
    N = np.zeros((24, 3), np.float32)
    M = np.zeros((24, 11), np.float32)
    L.pt_scene_get(P(V), P(I, C.c_uint32), P(N), P(M))
    assert np.array_equal(V, np.ctypeslib.as_array(OL.orc_scene_vertices(), shape=(48, 3)))
    assert np.array_equal(I, np.ctypeslib.as_array(OL.orc_scene_indices(), shape=(24, 3)))
    assert np.array_equal(N, np.ctypeslib.as_array(OL.orc_scene_normals(), shape=(24, 3)))


# ------------------------------------------------------------------ a7 ray casting
def test_trace_closest_matches_brute_force(gpu, oracle):
    o, d = _random_rays(200000, 1)
    prim, t, b1, b2 = gpu.trace_rays(o, d)
    m = 20000
    op, ot = _orc_trace(oracle, o[:m], d[:m])
    mism = (prim[:m] != op).sum()
    assert mism <= 2, mism  # index work: exact, bar rays that graze a shared edge within rounding
    same = (prim[:m] == op) & (op >= 0)
    assert np.abs(t[:m][same] - ot[same]).max() < 1e-5  # fp32, v_rcp (1 ulp) in the final division
    hit = prim >= 0
    assert ((b1 >= 0) & (b2 >= 0) & (b1 + b2 <= 1 + 1e-5))[hit].all()
    # golden hits
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    go = np.array([h["o"] for h in kat["hits"]], np.float32)
    gd = np.array([h["d"] for h in kat["hits"]], np.float32)
    gp, gt, gb1, gb2 = gpu.trace_rays(go, gd)
    assert gp.tolist() == [h["prim"] for h in kat["hits"]]
    for i, h in enumerate(kat["hits"]):
        if h["prim"] >= 0:
            assert gt[i] == pytest.approx(h["t"], rel=1e-5) and gb1[i] == pytest.approx(h["b1"], abs=1e-5)


def test_trace_camera_rays_and_any_hit(gpu, oracle):
    OL = oracle.lib()
    w, h = 160, 90
    eye, u, v, ww = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    OL.orc_camera_default(w, h, eye, u, v, ww)
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    dirs = np.zeros((w * h, 3), np.float32)
    for i, (x, y) in enumerate(zip(xs.ravel(), ys.ravel())):
        d = (C.c_float * 3)()
        OL.orc_camera_dir(int(x), int(y), (C.c_float * 2)(0, 0), w, h, u, v, ww, d)
        dirs[i] = list(d)
    org = np.tile(np.array(list(eye), np.float32), (w * h, 1))
    prim, t, _, _ = gpu.trace_rays(org, dirs)
    op, ot = _orc_trace(oracle, org, dirs)
    # pixel-centre rays of the middle column lie in the plane x = 0 and run exactly along shared
    # edges: either neighbour is a correct answer there, at the same distance
    differ = prim != op
    assert differ.mean() < 0.02 and ((prim >= 0) == (op >= 0)).all()
    assert np.abs(t - ot)[op >= 0].max() < 1e-5
    # open front: the outer columns see nothing (black side bars of the reference's renders)
    assert (prim.reshape(h, w)[:, :20] == -1).all() and (prim.reshape(h, w)[:, -20:] == -1).all()
    # shadow-ray semantics: occluded iff something lies in (0, tmax)
    o, d = _random_rays(20000, 5)
    tmax = np.random.default_rng(6).uniform(0.05, 2.5, len(o)).astype(np.float32)
    occ, _, _, _ = gpu.trace_rays(o, d, None, tmax, any_hit=True)
    cp, ct, _, _ = gpu.trace_rays(o, d)
    want = ((cp >= 0) & (ct < tmax)).astype(np.int32)
    edge = np.abs(ct - tmax) < 1e-5
    assert np.array_equal(occ[~edge], want[~edge])
    fp = C.POINTER(C.c_float)
    ow = np.array([OL.orc_trace_any(o[i].ctypes.data_as(fp), d[i].ctypes.data_as(fp), 0.0, float(tmax[i])) for i in range(4000)])
    assert (occ[:4000] != ow).sum() <= 1


def test_trace_synthetic_scene_matches_brute_force(gpu, oracle):
    """Deeper BVH out of HBM: the room plus 125 spheres of 1280 triangles (160k triangles)."""
    L, OL = gpu.lib(), oracle.lib()
    L.pt_scene_synthetic(42, 125, 3)
    try:
        info = gpu.scene_info()
        nt = info["triangles"]
        assert nt == 12 + 125 * 1280 and info["depth"] >= 5
        nv = 24 + 125 * 642
        V = np.zeros((nv, 3), np.float32)
        I = np.zeros((nt, 3), np.uint32)
        L.pt_scene_get(P(V), P(I, C.c_uint32), None, None)
        OL.orc_scene_set(P(V), nv, P(I, C.c_uint32), nt, None)
        o, d = _random_rays(400, 9)
        prim, t, _, _ = gpu.trace_rays(o, d)
        op, ot = _orc_trace(oracle, o, d)
        assert (prim != op).sum() == 0
        assert np.abs(t - ot)[op >= 0].max() < 1e-5
        # size-independent property on many rays: the closest hit is inside the closed box walls
        o, d = _random_rays(500000, 10)
        prim, t, b1, b2 = gpu.trace_rays(o, d)
        away = d[:, 1] < 0  # heading away from the open front
        assert (prim[away] >= 0).all()
        img = gpu.render_rgb(64, 36, 8)
        assert np.isfinite(img).all() and img.mean() > 0.01
        # a BVH this size runs staged and hands the stragglers to the fused kernel: same bits as either alone
        imgs = []
        L.pt_set_defer_shadow(0)
        for pipeline in (-1, 0, 1):
            L.pt_set_pipeline(pipeline)
            L.pt_set_frame(7)
            imgs.append(gpu.render_rgb(160, 90, 16))
        assert np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[1], imgs[2])
        L.pt_set_defer_shadow(1)
        L.pt_set_frame(7)
        assert np.allclose(gpu.render_rgb(160, 90, 16), imgs[2], rtol=2e-5, atol=1e-7)
    finally:
        OL.orc_scene_set(None, 0, None, 0, None)
        L.pt_scene_cornell()
        L.pt_set_pipeline(1)
        L.pt_set_defer_shadow(DEFER_DEFAULT)


def test_awkward_geometry_is_sound(gpu):
    """Degenerate triangles, slivers and millimetre triangles 50 units from the origin through the GPU
    traversal: every answer is consistent with a float64 ground truth (tests/geom_cases.py)."""
    import geom_cases
    L = gpu.lib()
    V, I, centre, rng = geom_cases.awkward_scene()
    L.pt_scene_triangles(P(V), len(V), P(I, C.c_uint32), len(I), None)
    try:
        o, d = geom_cases.awkward_rays(V, I, centre, rng)
        prim, t, b1, b2 = gpu.trace_rays(o, d)
        geom_cases.check_against_float64(V, I, o, d, prim, t, b1, b2)
        # any-hit: occluded exactly when the closest hit lies inside the interval (away from its ends)
        tmax = rng.uniform(0.2, 4.0, len(o)).astype(np.float32)
        occ, _, _, _ = gpu.trace_rays(o, d, None, tmax, any_hit=True)
        want = ((prim >= 0) & (t < tmax)).astype(np.int32)
        clear = np.abs(t - tmax) > 1e-3
        assert (occ[clear] >= want[clear]).all()  # an occluder behind the closest hit may also be found
        assert (occ[(prim < 0)] == 0).all()
    finally:
        L.pt_scene_cornell()


def test_parallelogram_items(gpu):
    """Quads given as two consecutive triangles are traversed as one item each: right primitive, that
    primitive's own barycentrics, for every diagonal / winding / corner order (tests/geom_cases.py), and the
    same render as with one item per triangle."""
    import geom_cases
    L = gpu.lib()
    V, I, centre, rng, n_quads = geom_cases.quad_scene()
    L.pt_scene_triangles(P(V), len(V), P(I, C.c_uint32), len(I), None)
    try:
        o, d = geom_cases.quad_rays(V, I, rng, 20000)
        prim, t, b1, b2 = gpu.trace_rays(o, d)
        info = gpu.scene_info()
        assert (info["bvh_bytes"] - 128 * info["nodes"]) // 64 == len(I) - n_quads
        geom_cases.check_against_float64(V, I, o, d, prim, t, b1, b2)
        L.pt_set_bvh_pairing(0)
        prim0, t0, b10, b20 = gpu.trace_rays(o, d)
        info = gpu.scene_info()
        assert (info["bvh_bytes"] - 128 * info["nodes"]) // 64 == len(I)
        same = prim == prim0
        assert same.mean() > 0.98
        hit = same & (prim >= 0)
        db = np.maximum(np.abs(b1[hit] - b10[hit]), np.abs(b2[hit] - b20[hit]))
        assert np.abs(t[hit] - t0[hit]).max() < 2e-3 and db.max() < 5e-3  # grazing rays 40 units out
        assert np.quantile(np.abs(t[hit] - t0[hit]), 0.999) < 2e-4 and np.quantile(db, 0.999) < 2e-4
    finally:
        L.pt_set_bvh_pairing(-1)
        L.pt_scene_cornell()
    # the built-in scene (twelve quads) renders the same image with and without pairing
    w, h, spp = 96, 54, 256
    imgs, rays = [], []
    for pairing in (1, 0):
        L.pt_set_bvh_pairing(pairing)
        L.pt_set_frame(0)
        L.pt_reset_ray_counts()
        imgs.append(gpu.render_rgb(w, h, spp))
        rays.append(sum(gpu.ray_counts()))
    L.pt_set_bvh_pairing(-1)
    # a path that starts exactly on a plane (0.75 % of rays: below the glass box) sees that plane or not by the
    # sign of a rounding error, which the two record layouts do not share; every other ray is the same ray
    assert abs(rays[0] - rays[1]) / rays[1] < 5e-3
    assert abs(imgs[0].mean() - imgs[1].mean()) / imgs[1].mean() < 0.01
    close = (np.abs(imgs[0] - imgs[1]) <= 2e-3 + 2e-2 * np.abs(imgs[1])).all(axis=2).mean()
    assert close > 0.5, close  # most pixels follow the very same paths; the rest are re-drawn


def test_obj_scene_renders_like_the_built_in_one(gpu, tmp_path):
    """The built-in scene saved as OBJ + MTL and loaded again renders the same image bit for bit, and a
    light set by hand to the reference's values changes nothing."""
    L = gpu.lib()
    L.pt_scene_cornell()
    L.pt_set_frame(0)
    want = gpu.render_rgb(96, 54, 64)
    path = str(tmp_path / "cornell.obj").encode()
    assert L.pt_scene_save_obj(path) == 0
    try:
        assert L.pt_scene_load_obj(path) == 24
        L.pt_set_frame(0)
        assert np.array_equal(gpu.render_rgb(96, 54, 64), want)
        f = lambda *v: np.array(v, np.float32)
        L.pt_scene_set_light(P(f(0, 0, 1.99)), P(f(15, 15, 15)), P(f(0, 0, -1)), P(f(0.3, 0.3)))
        L.pt_set_frame(0)
        assert np.array_equal(gpu.render_rgb(96, 54, 64), want)
        L.pt_scene_set_light(None, P(f(30, 30, 30)), None, None)  # twice the emission for next-event estimation
        L.pt_set_frame(0)
        brighter = gpu.render_rgb(96, 54, 64)
        assert 1.3 < brighter.mean() / want.mean() < 2.0  # the panel's own emission (hit by chance) is unchanged
    finally:
        L.pt_scene_cornell()


def test_watertight_closed_box(gpu):
    """No ray leaving an interior point away from the open front escapes (shared diagonals / edges)."""
    rng = np.random.default_rng(3)
    n = 1000000
    o = np.tile(np.array([[0.1, 0.2, 1.0]], np.float32), (n, 1))
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:, 1] = -np.abs(d[:, 1]) - 1e-3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    prim, _, _, _ = gpu.trace_rays(o, d)
    assert (prim >= 0).all()
    # rays aimed exactly at the diagonals and edges of the walls
    s = np.linspace(-1, 0.95, 2001, dtype=np.float32)  # stay off the rim of the open front (y = +1)
    tg = np.concatenate([np.stack([s, s, 0 * s], 1), np.stack([s, -np.ones_like(s), 1 + s], 1),
                         np.stack([-np.ones_like(s), s, 1 - s], 1), np.stack([np.ones_like(s), s, 1 + s], 1),
                         np.stack([s, -s, 2 + 0 * s], 1), np.stack([s, -np.ones_like(s), 0 * s], 1)])
    oo = np.tile(np.array([[0.123, 0.3, 1.1]], np.float32), (len(tg), 1))
    dd = tg - oo
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    prim, _, _, _ = gpu.trace_rays(oo, dd.astype(np.float32))
    assert (prim >= 0).all()


# ------------------------------------------------------------------ a8 shading step
def _gpu_shade(gpu, frame=1, max_depth=32):
    def shade(recs, prims, b1, b2):
        shadow = np.zeros(len(prims), np.int32)
        gpu.lib().pt_dbg_shade(len(prims), C.cast(recs, C.POINTER(gpu.PathRec)), P(prims, C.c_int), P(b1), P(b2),
                               frame, max_depth, P(shadow, C.c_int))
        return recs, shadow
    return shade


def test_shade_step_matches_oracle(gpu, oracle):
    cases = make_shade_cases(oracle, 6000, 33)
    frac = check_shade_against_oracle(oracle, cases, _gpu_shade(gpu))
    # fp32 with 1-ulp hardware rcp/sqrt/sin/cos against libm: 2e-3 relative on contributions; a
    # handful of cases flip a branch (random number within rounding of a probability)
    assert frac > 0.99, frac


@pytest.mark.parametrize("case", [0, 1])
def test_shade_step_means_match_quadrature(gpu, oracle, case):
    """The GPU shading step against mathematics, not against the oracle: 400 000 random states at a floor
    point; mean next-event contribution, mean throughput factor and mean total contribution against the
    float64 quadrature of optix_device.cu:484-658 in tests/math_pins.py, within 4.5 standard errors."""
    import math_pins as mp
    from test_oracle_golden import FLOOR_CASES, floor_case
    eye, point = FLOOR_CASES[case]
    hit, eye32, d32, n, v = floor_case(oracle, eye, point)  # the oracle only looks the primitive up
    e_lo = mp.expected_nee(np.array(point, np.float64), n, v)
    e_w, _ = mp.expected_throughput_factor(n, v)
    t0 = np.array([0.9, 0.7, 0.5])
    N = 400000
    recs = (HcRec * N)()
    rng = np.random.default_rng(12)
    seeds = rng.integers(0, 2 ** 63, N, dtype=np.uint64)
    for i in range(N):
        r = recs[i]
        r.throughput[:] = t0.tolist()
        r.origin[:] = eye32.tolist()
        r.direction[:] = d32.tolist()
        r.seed = int(seeds[i])
        r.rng_dim = 2
    prims = np.full(N, hit.prim, np.int32)
    b1 = np.full(N, hit.b1, np.float32)
    b2 = np.full(N, hit.b2, np.float32)
    out, shadow = _gpu_shade(gpu)(recs, prims, b1, b2)
    assert shadow.all()
    arr = np.frombuffer(out, dtype=np.uint8).reshape(N, C.sizeof(HcRec))
    f = arr[:, :48].copy().view(np.float32).reshape(N, 12).astype(np.float64)
    contrib, thr = f[:, 0:3], f[:, 3:6]  # the radiance slot returns radiance * throughput
    for name, samples, expect in (("throughput", thr, t0 * e_w), ("contribution", contrib, e_lo * t0 * e_w)):
        mean, sem = samples.mean(0), samples.std(0) / np.sqrt(N)
        assert (np.abs(mean - expect) < 4.5 * sem + 2e-4 * expect).all(), (name, mean, expect, sem)
        assert (sem < 0.03 * expect).all()
    # the next-event term alone: contribution / throughput, where the throughput is not tiny
    ok = thr.min(1) > 1e-3
    lo = contrib[ok] / thr[ok]
    mean, sem = lo.mean(0), lo.std(0) / np.sqrt(ok.sum())
    assert ok.mean() > 0.99 and (np.abs(mean - e_lo) < 4.5 * sem + 2e-3 * e_lo).all(), (mean, e_lo, sem)


def test_glass_and_emission_follow_their_formulas_on_the_gpu(gpu, oracle):
    """The GPU shading step at the glass box and at the light against closed forms (tests/test_oracle_golden.py:
    Schlick's reflection share, Snell's law, total internal reflection, the restart offset; emission weighted by
    the balance heuristic).  The oracle is only asked for the scene's tables."""
    from test_oracle_golden import glass_cases, check_glass_step, light_hit_cases, expected_light_hit
    t_in = np.array([0.9, 0.7, 0.5])
    shade = _gpu_shade(gpu)
    N = 6000
    seeds = np.random.default_rng(3).integers(0, 2 ** 63, N, dtype=np.uint64)
    for prim, point, n, ior, base, d, inside in glass_cases(oracle):
        recs = (HcRec * N)()
        eye = (point - 0.5 * d).astype(np.float32)
        for i in range(N):
            r = recs[i]
            r.throughput[:] = t_in.tolist()
            r.origin[:] = eye.tolist()
            r.direction[:] = d.astype(np.float32).tolist()
            r.seed, r.rng_dim = int(seeds[i]), 2
        out, _ = shade(recs, np.full(N, prim, np.int32), np.full(N, 0.3, np.float32), np.full(N, 0.3, np.float32))
        f = np.frombuffer(out, dtype=np.uint8).reshape(N, C.sizeof(HcRec))[:, :48].copy().view(np.float32).reshape(N, 12)
        check_glass_step([(f[i, 9:12], f[i, 6:9], f[i, 3:6]) for i in range(N)], n, ior, base, d, inside, point, t_in)
    cases = [(c, prev) for c in light_hit_cases(oracle) for prev in (None, 0.05, 0.8, 30.0)]
    recs = (HcRec * len(cases))()
    for r, ((prim, b1, b2, point, emission, org, d), prev) in zip(recs, cases):
        r.throughput[:] = [1.0, 1.0, 1.0]
        r.origin[:] = org.tolist()
        r.direction[:] = d.tolist()
        r.seed, r.depth, r.rng_dim = 12345, 1, 2
        r.prev_pdf_valid, r.prev_pdf_bsdf = (0, 0.0) if prev is None else (1, prev)
    out, _ = shade(recs, np.array([c[0][0] for c in cases], np.int32), np.array([c[0][1] for c in cases], np.float32),
                   np.array([c[0][2] for c in cases], np.float32))
    f = np.frombuffer(out, dtype=np.uint8).reshape(len(cases), C.sizeof(HcRec))[:, :48].copy().view(np.float32).reshape(-1, 12)
    for row, ((prim, b1, b2, point, emission, org, d), prev) in zip(f.astype(np.float64), cases):
        want = expected_light_hit(point, emission, org, d, prev)
        assert np.allclose(row[0:3] / row[3:6], want, rtol=1e-4), (row[0:3] / row[3:6], want)  # contribution / throughput after


def test_vmf_samples_on_the_gpu_follow_their_density(gpu):
    """vmf_sample on the device (through pt_dbg_guiding, one region, one lobe) against the analytic von
    Mises-Fisher cdf and a uniform azimuth, chi-square, for a broad, a medium and a sharp lobe; the mixture pdf
    of a single lobe against kappa / (4 pi sinh kappa) exp(kappa mu . omega)."""
    import math_pins as mp
    L = gpu.lib()
    rng = np.random.default_rng(8)
    mu = np.array([0.36, -0.48, 0.8])
    t1 = np.cross(mu, [0, 0, 1.0]); t1 /= np.linalg.norm(t1)
    t2 = np.cross(mu, t1)
    n_s = 200000
    try:
        for kappa in (0.5, 8.0, 60.0):
            regions = np.array([[-1, -1, 0, 1, 1, 2, 0, 1]], np.float32)
            lobes = np.array([[mu[0], mu[1], mu[2], kappa, 1.0, 1, 1, 1]], np.float32)
            L.guiding_set_grid_res(1, 1, 1)
            gpu.upload_snapshot(regions, lobes)
            pos = np.tile(np.array([[0.1, 0.2, 0.7]], np.float32), (n_s, 1))
            u3 = rng.random((n_s, 3)).astype(np.float32)
            wi = rng.normal(size=(n_s, 3)).astype(np.float32)
            wi /= np.linalg.norm(wi, axis=1, keepdims=True)
            reg, lobe = np.zeros(n_s, np.int32), np.zeros(n_s, np.int32)
            dirs = np.zeros((n_s, 3), np.float32)
            pd, pw = np.zeros(n_s, np.float32), np.zeros(n_s, np.float32)
            assert L.pt_dbg_guiding(n_s, P(pos), P(u3), P(wi), P(reg, C.c_int), P(lobe, C.c_int), P(dirs), P(pd), P(pw)) == 0
            assert (reg == 0).all() and (lobe == 0).all()
            w = dirs.astype(np.float64)
            assert np.allclose(np.linalg.norm(w, axis=1), 1.0, atol=1e-4)
            cdf = mp.vmf_cdf(w @ mu, kappa)
            phi = np.mod(np.arctan2(w @ t2, w @ t1), 2 * np.pi) / (2 * np.pi)
            for values, bins in ((cdf, 20), (phi, 12)):
                counts = np.histogram(values, bins=np.linspace(0, 1, bins + 1))[0]
                chi2 = ((counts - n_s / bins) ** 2 / (n_s / bins)).sum()
                assert chi2 < mp.chi2_threshold(bins - 1), (kappa, bins, chi2)
            exact = kappa / (2 * np.pi * (1 - np.exp(-2 * kappa))) * np.exp(kappa * (wi.astype(np.float64) @ mu - 1))
            assert np.allclose(pw, exact, rtol=2e-3, atol=1e-7)
    finally:
        L.guiding_set_enabled(0)


def test_depth_one_render_matches_quadrature(gpu):
    """The GPU's whole render path at max_depth = 1 against the same quadrature the oracle is pinned with."""
    from test_oracle_golden import DEPTH_ONE, check_depth_one_block
    gpu.lib().pt_set_frame(2)
    img = gpu.render_rgb(DEPTH_ONE["width"], DEPTH_ONE["height"], DEPTH_ONE["spp"], max_depth=1)
    check_depth_one_block(img)


def test_shade_miss_terminates(gpu):
    n = 8
    recs = (HcRec * n)()
    for r in recs:
        r.throughput[:] = [1, 1, 1]
        r.direction[:] = [0, 1, 0]
        r.seed = 5
    prims = np.full(n, -1, np.int32)
    z = np.zeros(n, np.float32)
    out, shadow = _gpu_shade(gpu)(recs, prims, z, z)
    for r in out:
        assert r.done == 1 and list(r.radiance) == [0, 0, 0] and r.seed == 5
    assert not shadow.any()


# ------------------------------------------------------------------ a13 / a14 guiding
def synthetic_snapshot(seed=1234, res=(4, 4, 4), lobes_per_region=4):
    rng = np.random.default_rng(seed)
    regions, lobes = [], []
    lo, hi = np.array([-1, -1, 0.0]), np.array([1, 1, 2.0])
    step = (hi - lo) / np.array(res)
    for z in range(res[2]):
        for y in range(res[1]):
            for x in range(res[0]):
                bmin = lo + step * np.array([x, y, z])
                regions.append(list(bmin) + list(bmin + step) + [len(lobes), lobes_per_region])
                for _ in range(lobes_per_region):
                    mu = rng.normal(size=3)
                    mu /= np.linalg.norm(mu)
                    lobes.append(list(mu) + [rng.uniform(1, 50), rng.uniform(0.05, 1.0), 1.0, 1.0, 1.0])
    return np.array(regions, np.float32), np.array(lobes, np.float32)


def test_guiding_lookup_and_vmf(gpu, oracle):
    L, OL = gpu.lib(), oracle.lib()
    regions, lobes = synthetic_snapshot()
    L.guiding_set_grid_res(8, 8, 8)
    gpu.upload_snapshot(regions, lobes)
    G, keep = oracle.make_guide(regions, lobes, (8, 8, 8))
    rng = np.random.default_rng(2)
    n = 4000
    pos = rng.uniform([-1.2, -1.2, -0.2], [1.2, 1.2, 2.2], (n, 3)).astype(np.float32)
    u3 = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    wi = rng.normal(size=(n, 3)).astype(np.float32)
    wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    region = np.zeros(n, np.int32)
    lobe = np.zeros(n, np.int32)
    gdir = np.zeros((n, 3), np.float32)
    pdf_dir = np.zeros(n, np.float32)
    pdf_wi = np.zeros(n, np.float32)
    assert L.pt_dbg_guiding(n, P(pos), P(u3), P(wi), P(region, C.c_int), P(lobe, C.c_int), P(gdir), P(pdf_dir), P(pdf_wi)) == 0
    bad_region = bad_lobe = 0
    for i in range(n):
        rid = OL.orc_guiding_region_id(C.byref(G), oracle.f3(pos[i]))
        bad_region += rid != region[i]
        if rid < 0 or rid != region[i]:
            continue
        R = G.regions[rid]
        Lb = C.cast(C.addressof(G.lobes.contents) + 32 * R.lobe_ofs, C.POINTER(oracle.Lobe))
        li = OL.orc_guiding_choose_lobe(C.byref(R), Lb, float(u3[i, 0]), -1)
        bad_lobe += li != lobe[i]
        want_pdf = OL.orc_guiding_mixture_pdf(C.byref(R), Lb, oracle.f3(wi[i]))
        assert pdf_wi[i] == pytest.approx(want_pdf, rel=2e-4, abs=1e-7)
        if li == lobe[i]:
            out = (C.c_float * 3)()
            OL.orc_vmf_sample(Lb[li].mu, Lb[li].kappa, (C.c_float * 2)(float(u3[i, 1]), float(u3[i, 2])), out)
            assert np.abs(gdir[i] - np.array(list(out))).max() < 2e-4
    assert bad_region == 0
    assert bad_lobe <= 2  # u * sum compared with a running sum: a tie within rounding may pick a neighbour
    L.guiding_set_enabled(0)


def test_guided_shade_step_matches_oracle(gpu, oracle):
    """Guided diffuse branch (optix_device.cu:596-658) with the field enabled."""
    L = gpu.lib()
    regions, lobes = synthetic_snapshot(seed=77)
    L.guiding_set_grid_res(4, 4, 4)
    gpu.upload_snapshot(regions, lobes)
    L.guiding_set_enabled(1)
    try:
        G, keep = oracle.make_guide(regions, lobes, (4, 4, 4))
        cases = make_shade_cases(oracle, 4000, 55)
        frac = check_shade_against_oracle(oracle, cases, _gpu_shade(gpu), guide=G, rtol=5e-3)
        assert frac > 0.985, frac
        img = gpu.render_rgb(48, 27, 16)
        assert np.isfinite(img).all()
    finally:
        L.guiding_set_enabled(0)


@pytest.mark.parametrize("lobe", [0, 1, 2, 3])
def test_guided_branch_on_the_gpu_matches_quadrature(gpu, oracle, lobe):
    """The GPU's guided diffuse branch (device lobe records, batched passes, hardware exp2 / log) against the float64
    integral of the reference's selection rule: mean throughput factor of 400 000 shading steps at a floor point whose
    region holds one lobe or three, 4.5 standard errors.  The oracle only looks the primitive up."""
    import math_pins as mp
    from test_oracle_golden import FLOOR_CASES, floor_case, GUIDE_PIN_LOBES, lobes_field, as_float32_lobes
    L = gpu.lib()
    eye, point = FLOOR_CASES[0]
    hit, eye32, d32, n, v = floor_case(oracle, eye, point)
    regions, lobes = lobes_field(GUIDE_PIN_LOBES[lobe])
    want = mp.expected_guided_throughput_factor(n, v, as_float32_lobes(lobes))
    L.guiding_set_grid_res(1, 1, 1)
    gpu.upload_snapshot(regions, lobes)
    L.guiding_set_enabled(1)
    try:
        t0 = np.array([0.9, 0.7, 0.5])
        N = 400000
        recs = (HcRec * N)()
        seeds = np.random.default_rng(21 + lobe).integers(0, 2 ** 63, N, dtype=np.uint64)
        for i in range(N):
            r = recs[i]
            r.throughput[:] = t0.tolist()
            r.origin[:] = eye32.tolist()
            r.direction[:] = d32.tolist()
            r.seed, r.rng_dim = int(seeds[i]), 2
        out, _ = _gpu_shade(gpu)(recs, np.full(N, hit.prim, np.int32), np.full(N, hit.b1, np.float32), np.full(N, hit.b2, np.float32))
        f = np.frombuffer(out, dtype=np.uint8).reshape(N, C.sizeof(HcRec))[:, :48].copy().view(np.float32).reshape(N, 12)
        thr = f[:, 3:6].astype(np.float64)
        mean, sem = thr.mean(0), thr.std(0) / np.sqrt(N)
        assert (np.abs(mean - t0 * want) < 4.5 * sem + 2e-4 * t0 * want).all(), (mean, t0 * want, sem)
        assert (sem < 0.02 * t0 * want).all()
    finally:
        L.guiding_set_enabled(0)
        L.guiding_set_grid_res(1, 1, 1)
        gpu.upload_snapshot(np.zeros((0, 8), np.float32), np.zeros((0, 8), np.float32))


@pytest.mark.parametrize("pipeline", [0, 1])
def test_guided_render_parity(gpu, oracle, pipeline):
    """Whole guided renders (field on, seeded synthetic snapshot) against the oracle's guided render:
    same estimator including the reference's biased guided branch, so the images agree statistically
    and differ measurably from the unguided image."""
    L = gpu.lib()
    regions, lobes = synthetic_snapshot(seed=1234)
    w, h, spp = 64, 48, 512
    G, keep = oracle.make_guide(regions, lobes, (4, 4, 4))
    refs = [oracle.render(w, h, spp, frame=f, guide=G) for f in (1, 2, 3)]
    ref, orays = refs[0][0], refs[0][3]
    plain = oracle.render(w, h, spp, frame=1)[0]
    L.guiding_set_grid_res(4, 4, 4)
    gpu.upload_snapshot(regions, lobes)
    L.guiding_set_enabled(1)
    L.pt_set_pipeline(pipeline)
    try:
        imgs = []
        for f in (30, 31, 32, 33):
            L.pt_set_frame(f)
            L.pt_reset_ray_counts()
            imgs.append(gpu.render_rgb(w, h, spp))
            rr, rs = gpu.ray_counts()
        img = imgs[0]
    finally:
        L.guiding_set_enabled(0)
        L.pt_set_pipeline(1)
    # image means: the guided estimator has a heavy tail (weights 1 / pdf of a sharp lobe), so one frame's mean moves
    # by about a per cent between frames on either side (measured: oracle 0.1249 / 0.1258 / 0.1261, GPU 0.1254 ...
    # 0.1276); several frames of each are compared, against the spread they show
    m_gpu, m_ref = np.array([x.mean() for x in imgs]), np.array([r[0].mean() for r in refs])
    sigma = np.sqrt(m_gpu.var(ddof=1) / len(m_gpu) + m_ref.var(ddof=1) / len(m_ref))
    assert abs(m_gpu.mean() - m_ref.mean()) < 3 * sigma + 0.005 * m_ref.mean(), (m_gpu, m_ref)
    assert abs(img.mean() - ref.mean()) / ref.mean() < 0.04
    assert rel_rmse_trimmed(img, ref) < 0.2  # two independent 512-spp estimates: 0.14-0.15 in the oracle
    assert abs((rr + rs) - sum(orays)) / sum(orays) < 0.02
    # the guided estimator is not the unguided one (the reference's guided branch is biased)
    assert abs(ref.mean() - plain.mean()) / plain.mean() > 0.03
    assert abs(img.mean() - plain.mean()) / plain.mean() > 0.03


def test_guiding_loop_through_the_helper(gpu):
    """Render with the training ring on, train the stand-in field from the ring through the multi-GPU helper
    (one rank here), upload the snapshot, render guided: the loop of src/main.rs:496-540."""
    from ptamd import sharded
    L = gpu.lib()
    L.guiding_set_train_buffer_size(1 << 14)
    L.guiding_set_grid_res(4, 4, 4)
    try:
        L.pt_set_frame(0)
        plain = gpu.render_rgb(48, 36, 64)  # new resolution -> renderer with a ring
        rec = gpu.train_samples()
        assert 0.9 * (1 << 14) < len(rec) <= (1 << 14) and (rec[:, 9].view(np.uint32) == 0).all()
        g = sharded.ShardedGuiding(gpu, (-1, -1, 0), (1, 1, 2))
        regions, lobes = g.process_batch()
        g.close()
        assert len(regions) >= 1 and len(lobes) >= len(regions) and len(gpu.train_samples()) == 0
        L.guiding_set_enabled(1)
        L.pt_set_frame(0)
        guided = gpu.render_rgb(48, 36, 64)
        assert np.isfinite(guided).all() and not np.array_equal(guided, plain)
        # the reference's guided branch is biased (it multiplies the throughput by the lobe colour, < 1 here):
        # darker, not broken
        assert 0.2 < guided.mean() / plain.mean() < 1.5
    finally:
        L.guiding_set_enabled(0)
        L.guiding_set_train_buffer_size(0)
        L.guiding_set_grid_res(1, 1, 1)
        gpu.upload_snapshot(np.zeros((0, 8), np.float32), np.zeros((0, 8), np.float32))
        gpu.render_rgb(44, 28, 1)  # a renderer without a ring for the tests that follow


def test_training_statistics_binned_on_the_gpu(gpu):
    """pt_guiding_bin_train_samples_device against tests/binning.py on the records guiding_map_train_samples
    returns for the same ring: the same int64 table, every word; and the one-call update (bin, copy 24 KB, update,
    reset) trains the field the record-by-record path of src/guiding.rs:194-236 trains."""
    import torch
    import binning
    from ptamd import sharded
    L = gpu.lib()
    L.guiding_set_train_buffer_size(1 << 14)
    a, b = gpu.GuidingField((-1, -1, 0), (1, 1, 2)), gpu.GuidingField((-1, -1, 0), (1, 1, 2))
    try:
        for batch in range(2):  # the second batch is assigned by axes the first one moved
            L.pt_set_frame(40 + batch)
            gpu.render_rgb(52, 36, 32)
            rec = gpu.train_samples()
            assert len(rec) > 0.9 * (1 << 14)
            want = binning.bin_records(rec, *b.layout())
            table = torch.zeros(want.size, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            b.bin_ring_device(table.data_ptr())
            got = table.cpu().numpy()
            assert 0.9 * len(rec) < got[7::8].sum() <= len(rec)  # records of weight zero or outside the field train nothing
            assert np.array_equal(got, want)
            a.update(rec)
            b.update_from_ring()
            assert len(gpu.train_samples()) == 0  # the ring was emptied
            (ra, la), (rb, lb) = a.snapshot(), b.snapshot()
            assert np.array_equal(ra, rb) and la.shape == lb.shape
            # sums in 2^-24 fixed point against doubles; which vertices a render leaves in the ring differs from run to
            # run, and a lobe that a few nearly parallel directions trained has a concentration that magnifies the last
            # bits of its mean length (kappa = r (3 - r^2) / (1 - r^2)): that column gets a tolerance of its own
            dev = np.abs(la - lb)
            assert np.allclose(la[:, 0:3], lb[:, 0:3], rtol=0, atol=2e-4), dev[:, 0:3].max()  # (a lobe of little weight: 6e-5 seen)
            assert np.allclose(la[:, 3], lb[:, 3], rtol=2e-3, atol=1e-4), (dev[:, 3] / np.maximum(np.abs(lb[:, 3]), 1e-3)).max()
            assert np.allclose(la[:, 4:8], lb[:, 4:8], rtol=1e-4, atol=2e-6), dev[:, 4:8].max()
        # the multi-GPU helper with one rank: the same calls, no collective
        gpu.render_rgb(52, 36, 8)
        g = sharded.BinnedGuiding(gpu, (-1, -1, 0), (1, 1, 2))
        regions, lobes = g.process_batch()
        g.close()
        assert len(regions) == 64 and len(lobes) >= 64 and len(gpu.train_samples()) == 0
    finally:
        a.close()
        b.close()
        L.guiding_set_train_buffer_size(0)
        gpu.upload_snapshot(np.zeros((0, 8), np.float32), np.zeros((0, 8), np.float32))
        gpu.render_rgb(44, 28, 1)  # a renderer without a ring for the tests that follow


def test_train_sample_ring(gpu):
    """optix_device.cu:431-441 + optix_wrapper.cpp:1270-1294.  Vertices go to hashed ring slots (no shared
    counter in the render's inner loop), so a render with several times more vertices than slots fills
    nearly all of them; the mapped records are the written slots only, never more than the ring holds."""
    L = gpu.lib()
    L.guiding_set_train_buffer_size(4096)
    try:
        gpu.render_rgb(40, 30, 4)  # new resolution -> new renderer -> ring allocated
        ptr = C.POINTER(gpu.TrainSample)()
        cnt = C.c_uint32()
        L.guiding_map_train_samples(C.byref(ptr), C.byref(cnt))
        assert 0.97 * 4096 < cnt.value <= 4096  # about 20 000 opaque hits happened: e^-5 of the slots stay empty
        recs = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(cnt.value, 10)).copy()
        flags = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(cnt.value, 10))[:, 9]
        assert (flags == 0).all()  # is_delta: only non-delta vertices are captured
        pos, din, thr = recs[:, 0:3], recs[:, 3:6], recs[:, 6:9]
        assert (np.abs(pos[:, 0]) <= 1.001).all() and (pos[:, 2] >= -1e-3).all() and (pos[:, 2] <= 2.001).all()
        assert np.abs(np.linalg.norm(din, axis=1) - 1).max() < 1e-3
        assert (thr >= 0).all() and np.isfinite(thr).all()
        L.guiding_reset_train_write_idx()
        L.guiding_map_train_samples(C.byref(ptr), C.byref(cnt))
        assert cnt.value == 0
        # a render with fewer vertices than slots: every mapped record is a real vertex, none invented
        L.pt_reset_ray_counts()
        gpu.render_rgb(40, 30, 1, max_depth=1)
        rr, _ = gpu.ray_counts()
        L.guiding_map_train_samples(C.byref(ptr), C.byref(cnt))
        assert 0 < cnt.value <= rr
        few = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(cnt.value, 10))
        assert np.abs(np.linalg.norm(few[:, 3:6], axis=1) - 1).max() < 1e-3
        for pipeline in (0, 1):  # both pipelines capture the same number of vertices
            L.pt_set_pipeline(pipeline)
            L.guiding_reset_train_write_idx()
            L.pt_set_frame(8)
            gpu.render_rgb(40, 30, 2)
            L.guiding_map_train_samples(C.byref(ptr), C.byref(cnt))
            counts = counts + [cnt.value] if pipeline else [cnt.value]
        assert abs(counts[0] - counts[1]) <= 0.02 * counts[0]  # same vertices, hash collisions differ in who wins
    finally:
        L.guiding_set_train_buffer_size(0)
        L.pt_set_pipeline(1)
        gpu.render_rgb(41, 30, 1)  # drop the renderer that owns the ring


def test_train_capture_is_the_oracles_set_of_vertices(gpu, oracle):
    """optix_device.cu:431-441 against the oracle, record by record.  The reference appends every non-delta
    vertex {position, -direction, throughput on arrival}; the device stores the same records at hashed slots.
    With a ring a thousand times larger than the number of vertices (no two share a slot) the mapped records must
    be the oracle's set: same count up to the paths that part ways at a grazing edge, and nearly every oracle
    record has a device twin within 1e-4."""
    from scipy.spatial import cKDTree
    L = gpu.lib()
    w, h, spp = 16, 12, 4
    got = []
    oracle.render(w, h, spp, frame=1, mode=oracle.STRICT, train=got)
    ref = got[0]
    assert len(ref) > 1500
    L.guiding_set_train_buffer_size(1 << 22)
    try:
        for pipeline in (0, 1):
            L.pt_set_pipeline(pipeline)
            L.pt_set_mode(gpu.MODE_STRICT)
            L.pt_set_frame(0)
            gpu.render_rgb(w, h, spp)  # new resolution -> renderer with the ring
            rec = gpu.train_samples()
            L.guiding_reset_train_write_idx()
            assert (rec[:, 9].view(np.uint32) == 0).all()
            assert abs(len(rec) - len(ref)) <= 0.03 * len(ref), (len(rec), len(ref))
            dist, _ = cKDTree(rec[:, :9].astype(np.float64)).query(ref.astype(np.float64))
            assert (dist < 1e-4).mean() > 0.9, (dist < 1e-4).mean()
            back, _ = cKDTree(ref.astype(np.float64)).query(rec[:, :9].astype(np.float64))
            assert (back < 1e-4).mean() > 0.9  # and the device invents none
    finally:
        L.pt_set_mode(gpu.MODE_CONVERGE)
        L.pt_set_pipeline(1)
        L.guiding_set_train_buffer_size(0)
        gpu.render_rgb(17, 12, 1)  # drop the renderer that owns the ring


# ------------------------------------------------------------------ a6 / a11 whole renders
@pytest.mark.parametrize("pipeline", [0, 1])
def test_strict_render_matches_golden(gpu, pipeline):
    """One 4-sample launch with the reference's seeds, pixel by pixel against the oracle image."""
    L = gpu.lib()
    g = np.load(os.path.join(GOLD, "strict_32x32_spp4_frame1.npz"))
    L.pt_set_pipeline(pipeline)
    L.pt_set_mode(gpu.MODE_STRICT)
    L.pt_set_frame(0)
    L.pt_reset_ray_counts()
    img = gpu.render_rgb(32, 32, 4)
    close = (np.abs(img - g["rgb"]) <= 2e-3 + 2e-3 * np.abs(g["rgb"])).all(axis=2).mean()
    assert close > 0.93, close  # measured 0.947; the rest are the rays that start on a plane (next test)
    rr, rs = gpu.ray_counts()
    assert abs(rr - g["rays"][0]) / g["rays"][0] < 0.02 and abs(rs - g["rays"][1]) / g["rays"][1] < 0.02
    # replay semantics: 16 spp strict = 4 repeats of the same launch
    L.pt_set_frame(0)
    img16 = gpu.render_rgb(32, 32, 16)
    assert np.allclose(img16, img, rtol=1e-5, atol=1e-6)
    # the mosaic of the same frame is the CFA channel of the same image
    L.pt_set_frame(0)
    raw = gpu.render_bayer(32, 32, 4, pattern=0)
    ch = np.array([[(x & 1) + (y & 1) for x in range(32)] for y in range(32)])  # RGGB: 0 / 1 / 2
    want = np.maximum(np.take_along_axis(img, ch[..., None], 2)[..., 0], 0)
    assert np.array_equal(raw, want)


def test_strict_mismatches_are_the_rays_that_start_on_a_plane(gpu, oracle):
    """Why a strict-mode render is not the oracle's in every pixel.  The reference pushes a new origin 1e-3 off
    the surface it leaves, which puts a path leaving the glass box through its bottom face (z = 0.001) exactly
    on the floor; whether that ray sees the floor is the sign of a rounding error, different in any two
    implementations.  The oracle marks the pixels where one of its rays starts in a triangle's plane: away from
    them the device agrees in (nearly) every pixel, and the pixels that disagree are (nearly) all marked."""
    L = gpu.lib()
    w, h, spp = 96, 72, 4
    marks = []
    ref = oracle.render(w, h, spp, frame=1, mode=oracle.STRICT, on_plane=marks)[0]
    marked = marks[0].astype(bool)
    assert 0.01 < marked.mean() < 0.3
    for pipeline in (0, 1):
        L.pt_set_pipeline(pipeline)
        L.pt_set_mode(gpu.MODE_STRICT)
        L.pt_set_frame(0)
        img = gpu.render_rgb(w, h, spp)
        close = (np.abs(img - ref) <= 2e-3 + 2e-3 * np.abs(ref)).all(axis=2)
        assert close[~marked].mean() > 0.985, close[~marked].mean()   # the rest: a grazing edge, a random number
        assert marked[~close].mean() > 0.7, marked[~close].mean()     # on a probability (branch flips)
        assert close[marked].mean() < close[~marked].mean()
    L.pt_set_mode(gpu.MODE_CONVERGE)
    L.pt_set_pipeline(1)


def test_pipelines_agree_and_are_deterministic(gpu):
    """The staged kernels, the fused kernel launch after launch and the fused kernel as one launch (the default:
    its lanes take slot after slot from a shared counter, so which wave carries which slot differs from run to run)
    render the same bits: the kernels are compiled with products and sums fused per source expression only, so they
    round every operation alike and take the same paths, and every form adds a slot's samples in sample order.
    With shadow rays queued per wave and traced 64 at a time (pt_set_defer_shadow(1)) the paths and the terms per
    pixel are the same and the order of additions is not: run-to-run identical, equal to rounding."""
    L = gpu.lib()
    assert L.pt_get_one_launch() == 1 and L.pt_get_defer_shadow() == DEFER_DEFAULT == -1

    def render(w, h, spp):
        L.pt_set_frame(4)
        L.pt_reset_ray_counts()
        img = gpu.render_rgb(w, h, spp)
        return img, gpu.ray_counts()

    try:
        L.pt_set_pipeline(0)
        staged, rays = render(96, 54, 32)
        staged_big, rays_big = render(1280, 720, 64)  # enough slots that every wave refills many times; two chunks
        L.pt_set_pipeline(1)
        assert np.array_equal(render(96, 54, 32)[0], staged)  # the default for this scene: shadow rays in place
        for defer in (0, 1):
            L.pt_set_defer_shadow(defer)
            for one_launch in (1, 0):
                L.pt_set_one_launch(one_launch)
                for _ in range(2):
                    img, r = render(96, 54, 32)
                    big, rb = render(1280, 720, 64)
                    assert r == rays and rb == rays_big
                    if defer == 0:
                        assert np.array_equal(img, staged) and np.array_equal(big, staged_big)
                    else:
                        assert np.allclose(img, staged, rtol=2e-5, atol=1e-7) and np.allclose(big, staged_big, rtol=2e-5, atol=1e-7)
                        if _ == 0:
                            first = (img, big)
                        else:
                            assert np.array_equal(img, first[0]) and np.array_equal(big, first[1])
        # adaptive renders: every round is one launch that picks the slots up where the last round left them (state
        # fetched 32 slots at a time into LDS) -- the bits and per-pixel counts of the staged kernels and of the fused
        # kernel launch after launch, with one stream per pixel and with several chunks
        L.pt_set_defer_shadow(DEFER_DEFAULT)
        for chunks in (1, 0):
            L.pt_set_adaptive_chunks(chunks)
            results = []
            for pipeline, one_launch in ((0, 1), (1, 1), (1, 0)):
                L.pt_set_pipeline(pipeline)
                L.pt_set_one_launch(one_launch)
                L.pt_set_frame(4)
                img = gpu.render_rgb(200, 120, 256, noise_threshold=1.5, min_adaptive_samples=32)
                results.append((img, gpu.sample_counts(200, 120)[0]))
            assert len(np.unique(results[0][1])) > 2  # it adapts
            for img, counts in results[1:]:
                assert np.array_equal(img, results[0][0]) and np.array_equal(counts, results[0][1])
    finally:
        L.pt_set_pipeline(1)
        L.pt_set_one_launch(1)
        L.pt_set_adaptive_chunks(0)
        L.pt_set_defer_shadow(DEFER_DEFAULT)


@pytest.mark.parametrize("spheres,subdiv,tris", [(1, 0, 32), (1, 1, 92), (2, 2, 652), (5, 2, 1612)])
def test_pipelines_agree_on_trees_of_several_nodes(gpu, spheres, subdiv, tris):
    """The same agreement on scenes whose tree is more than one node -- spheres in the room: 32 and 92 triangles (wide
    tree in LDS when shadow rays are traced in place, the default for them), 652 and 1 612 (packed tree from device
    memory; the larger one above the size where shadow rays are queued by default) -- where the fused kernels run the general traversal loop: staged kernels, fused launch after
    launch and fused as one launch give the same bits with shadow rays traced in place, the same image to rounding
    with the per-wave queue, the same ray counts always, and adaptive renders the same bits and counts."""
    L = gpu.lib()
    try:
        L.pt_scene_synthetic(42, spheres, subdiv)
        assert gpu.scene_info()["triangles"] == tris

        def render(w, h, spp, **kw):
            L.pt_set_frame(6)
            L.pt_reset_ray_counts()
            img = gpu.render_rgb(w, h, spp, **kw)
            return img, gpu.ray_counts()

        L.pt_set_pipeline(0)
        staged, rays = render(320, 180, 32)
        staged_ad, _ = render(160, 90, 128, noise_threshold=1.5, min_adaptive_samples=32)
        counts_ad = gpu.sample_counts(160, 90)[0]
        assert len(np.unique(counts_ad)) > 2
        L.pt_set_pipeline(1)
        img, r = render(320, 180, 32)  # the defaults for this scene
        assert r == rays and np.allclose(img, staged, rtol=2e-5, atol=1e-7)
        for defer in (0, 1):
            L.pt_set_defer_shadow(defer)
            for one_launch in (1, 0):
                L.pt_set_one_launch(one_launch)
                img, r = render(320, 180, 32)
                assert r == rays, (defer, one_launch)
                if defer == 0:
                    assert np.array_equal(img, staged), (defer, one_launch)
                else:
                    assert np.allclose(img, staged, rtol=2e-5, atol=1e-7), (defer, one_launch)
                    assert np.array_equal(img, render(320, 180, 32)[0])
                ad, _ = render(160, 90, 128, noise_threshold=1.5, min_adaptive_samples=32)
                assert np.array_equal(ad, staged_ad) and np.array_equal(gpu.sample_counts(160, 90)[0], counts_ad), (defer, one_launch)
        # ... and with a guiding field switched on (the guided instantiations of the same kernels)
        regions, lobes = synthetic_snapshot(seed=77)
        L.guiding_set_grid_res(4, 4, 4)
        gpu.upload_snapshot(regions, lobes)
        L.guiding_set_enabled(1)
        L.pt_set_one_launch(1)
        L.pt_set_pipeline(0)
        guided_staged, guided_rays = render(320, 180, 16)
        assert guided_rays != rays and not np.array_equal(guided_staged, staged)
        L.pt_set_pipeline(1)
        for defer in (0, 1):
            L.pt_set_defer_shadow(defer)
            img, r = render(320, 180, 16)
            assert r == guided_rays
            assert np.array_equal(img, guided_staged) if defer == 0 else np.allclose(img, guided_staged, rtol=2e-5, atol=1e-7)
    finally:
        L.guiding_set_enabled(0)
        L.guiding_set_grid_res(1, 1, 1)
        gpu.upload_snapshot(np.zeros((0, 8), np.float32), np.zeros((0, 8), np.float32))
        L.pt_set_pipeline(1)
        L.pt_set_one_launch(1)
        L.pt_set_defer_shadow(DEFER_DEFAULT)
        L.pt_scene_cornell()


@pytest.mark.parametrize("pipeline", [0, 1])
def test_converged_image_parity(gpu, pipeline):
    """Statistical parity with the high-spp oracle image: relRMSE falls as 1/sqrt(spp) down to the
    reference image's own noise, and the image mean is unbiased (3 sigma)."""
    L = gpu.lib()
    L.pt_set_pipeline(pipeline)
    ref = np.load(os.path.join(GOLD, "converge_96x54_spp4096_frame1.npz"))["rgb"]
    errs = {}
    for spp in (64, 1024):
        L.pt_set_frame(10)  # frame 11: independent of the oracle's frame-1 samples
        img = gpu.render_rgb(96, 54, spp)
        errs[spp] = rel_rmse(img, ref)
        if spp == 1024:
            final = img
    assert errs[1024] < 0.065, errs  # measured 0.047 (the 4096-spp reference image's own noise is 0.029)
    assert 2.0 < errs[64] / errs[1024] < 6.0, errs  # ideal 4
    # unbiased mean per channel: sigma of the image-mean estimator from two independent frames
    L.pt_set_frame(20)
    other = gpu.render_rgb(96, 54, 1024)
    for c in range(3):
        sigma = abs(final[..., c].mean() - other[..., c].mean()) / np.sqrt(2) + 1e-4 * ref[..., c].mean() + \
            np.std(final[..., c] - other[..., c]) / np.sqrt(2 * final[..., c].size)
        assert abs(final[..., c].mean() - ref[..., c].mean()) < 3 * sigma + 0.0015 * ref[..., c].mean()


def test_short_paths_match_the_oracle(gpu, oracle):
    """max_depth 4 puts the roulette of optix_device.cu:660-672 (last two depths, with the bias of adding
    radiance x throughput after the division) on bounces that carry energy: image means against the oracle's."""
    L = gpu.lib()
    L.pt_set_pipeline(-1)
    want = np.mean([oracle.render(96, 54, 512, max_depth=4, frame=f)[0].astype(np.float64) for f in (1, 2)], 0)
    L.pt_set_frame(30)
    got = [gpu.render_rgb(96, 54, 1024, max_depth=4).astype(np.float64) for _ in range(2)]
    for c in range(3):
        sigma = abs(got[0][..., c].mean() - got[1][..., c].mean()) / np.sqrt(2) + 1e-4 * want[..., c].mean()
        assert abs(0.5 * (got[0][..., c].mean() + got[1][..., c].mean()) - want[..., c].mean()) < 3 * sigma + 0.004 * want[..., c].mean()
    assert rel_rmse(got[0].astype(np.float32), want.astype(np.float32)) < 0.15


def test_bayer_converged_parity(gpu):
    L = gpu.lib()
    g = np.load(os.path.join(GOLD, "converge_64x64_spp1024_frame2.npz"))
    L.pt_set_frame(1)  # -> frame 2, the frame of the reference's Bayer call (src/main.rs:512-526)
    raw = gpu.render_bayer(64, 64, 1024, pattern=0)
    assert raw.min() >= 0
    assert rel_rmse(raw, g["bayer"]) < 0.085  # measured 0.063
    assert abs(raw.mean() - g["bayer"].mean()) / g["bayer"].mean() < 0.004  # measured 0.0006


def test_sample_chunks_sum_like_ranks(gpu):
    """Sample sharding: chunks [0,8) and [8,16) accumulated on one GPU are what two ranks reduce to;
    chunk [0,8) alone is the plain 8-spp render."""
    import torch
    L = gpu.lib()
    w, h = 64, 36
    dev = torch.device("cuda:0")
    sums = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu.render_accumulate_device(sums, w, h, 0, 8, 32, 3)
    first = sums.clone()
    L.pt_set_frame(2)
    plain = gpu.render_rgb(w, h, 8)  # frame 3
    assert np.allclose(first.cpu().numpy().reshape(h, w, 3) / 8, plain, rtol=1e-6, atol=1e-7)
    gpu.render_accumulate_device(sums, w, h, 8, 8, 32, 3)
    both = sums.cpu().numpy().reshape(h, w, 3)
    second = both - first.cpu().numpy().reshape(h, w, 3)
    assert not np.allclose(second, first.cpu().numpy().reshape(h, w, 3))
    assert abs(second.mean() - first.cpu().numpy().mean()) / second.mean() < 0.08
    # mosaic on the device from the summed image
    raw = torch.zeros(w * h, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    assert L.pt_mosaic_device(sums.data_ptr(), w, h, 1, 16.0, raw.data_ptr()) == 0
    r = raw.cpu().numpy().reshape(h, w)
    assert r[0, 0] == pytest.approx(max(both[0, 0, 2] / 16, 0), rel=1e-6)  # BGGR: (0,0) is blue


def test_in_library_shards_equal_the_ranks_they_stand_for(gpu):
    """pt_set_device_count(n): the reference's own entry point renders n sample shards side by side (here all
    on one GPU) and adds them in shard order.  The image is, bit for bit, what n ranks produce through
    pt_render_accumulate_device with the same sample ranges, summed in rank order and divided by the count;
    the mosaic entry point and the ray totals follow; one shard is the plain render."""
    import torch
    L = gpu.lib()
    w, h, spp, n = 96, 54, 100, 3
    dev = torch.device("cuda:0")
    try:
        L.pt_set_device_count(n)
        L.pt_set_frame(4)
        L.pt_reset_ray_counts()
        img = gpu.render_rgb(w, h, spp)  # frame 5
        rays_sharded = gpu.ray_counts()
        L.pt_set_frame(4)
        raw = gpu.render_bayer(w, h, spp, pattern=2)
        L.pt_set_device_count(1)
        total = None
        L.pt_reset_ray_counts()
        for d in range(n):
            b, c = C.c_int(), C.c_int()
            L.pt_shard_range(spp, n, d, C.byref(b), C.byref(c))
            part = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            gpu.render_accumulate_device(part, w, h, b.value, c.value, 32, 5)
            part = part.cpu().numpy()
            total = part if total is None else total + part
        assert gpu.ray_counts() == rays_sharded
        expect = (total * (np.float32(1.0) / np.float32(spp))).reshape(h, w, 3)
        assert np.array_equal(img.view(np.uint32), expect.view(np.uint32))
        ys, xs = np.mgrid[0:h, 0:w]
        ch = np.where(ys % 2 == 0, np.where(xs % 2 == 0, 1, 0), np.where(xs % 2 == 0, 2, 1))  # GRBG
        assert np.array_equal(raw, np.maximum(np.take_along_axis(expect, ch[..., None], axis=2)[..., 0], 0))
        # one shard: the plain render; more shards than samples: one sample each
        L.pt_set_frame(4)
        plain = gpu.render_rgb(w, h, spp)
        L.pt_set_device_count(8)
        L.pt_set_frame(4)
        few = gpu.render_rgb(w, h, 3)
        assert np.isfinite(few).all() and abs(few.mean() - plain.mean()) / plain.mean() < 0.2
        # adaptive renders shard too, with ONE noise test on the statistics of all shards together: the threshold and
        # min_adaptive_samples mean what they mean on one GPU (the black bars stop at 32 samples in all, not at 32 per
        # shard), counts follow the round schedule of the whole render, and the same pixels stop as in the one-shard
        # render, up to the noise of different samples
        L.pt_set_device_count(1)
        L.pt_set_frame(4)
        one = gpu.render_rgb(w, h, 256, noise_threshold=0.5, min_adaptive_samples=32)
        c_one, _ = gpu.sample_counts(w, h)
        for shards in (2, 3):
            L.pt_set_device_count(shards)
            L.pt_set_frame(4)
            ada = gpu.render_rgb(w, h, 256, noise_threshold=0.5, min_adaptive_samples=32)
            c_ada, rounds = gpu.sample_counts(w, h)
            assert np.isfinite(ada).all() and abs(ada.mean() - one.mean()) / one.mean() < 0.05
            # a shard takes its proportional part of every round, rounded up: 16 + 16, or 11 + 11 + 11, at the first test
            first = 32 if shards == 2 else 33
            assert c_ada.min() == first and c_ada.max() == 256 and rounds == 4
            assert (c_ada[:, 0] == first).all() and (shards != 2 or set(np.unique(c_ada)) <= {32, 64, 128, 256})
            assert (c_ada == c_one).mean() > 0.7, (shards, (c_ada == c_one).mean())
            L.pt_set_frame(4)
            again = gpu.render_rgb(w, h, 256, noise_threshold=0.5, min_adaptive_samples=32)
            assert np.array_equal(ada, again)  # whichever shard's thread arrives first
    finally:
        L.pt_set_device_count(1)


def test_boundary_utilities(gpu):
    """Pinned host buffers (optix_wrapper.cpp:1132-1164), synchronize, the explicit context API
    (927-976) and the four Bayer patterns (optix_device.cu:902-912)."""
    L = gpu.lib()
    w, h = 24, 16
    nbytes = w * h * 3 * 4
    ptr = L.optix_alloc_host(nbytes)
    assert ptr
    buf = (C.c_float * (w * h * 3)).from_address(ptr)
    L.pt_set_mode(gpu.MODE_STRICT)
    try:
        L.pt_set_frame(0)
        assert L.optix_render_rgb(w, h, 4, 32, C.c_float(0.0), 32, C.cast(ptr, C.POINTER(C.c_float))) == 0
        L.optix_synchronize()
        pinned = np.frombuffer(buf, np.float32).reshape(h, w, 3).copy()
        L.pt_set_frame(0)
        img = gpu.render_rgb(w, h, 4)
        assert np.array_equal(pinned, img)
        # pattern -> channel at (x&1, y&1): RGGB, BGGR, GRBG, GBRG
        tiles = {0: [[0, 1], [1, 2]], 1: [[2, 1], [1, 0]], 2: [[1, 0], [2, 1]], 3: [[1, 2], [0, 1]]}
        for pattern, tile in tiles.items():
            L.pt_set_frame(0)
            raw = gpu.render_bayer(w, h, 4, pattern=pattern)
            ch = np.array([[tile[y & 1][x & 1] for x in range(w)] for y in range(h)])
            want = np.maximum(np.take_along_axis(img, ch[..., None], 2)[..., 0], 0)
            assert np.array_equal(raw, want), pattern
    finally:
        L.optix_free_host(ptr)
        L.pt_set_mode(gpu.MODE_CONVERGE)
    L.optix_free_host(None)  # freeing NULL is a no-op
    ctx = L.optix_ctx_create(16, 8)
    assert ctx
    L.optix_ctx_set_bayer(ctx, 2)
    L.optix_ctx_set_camera(ctx, 0.0, 3.0, 1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 60.0)
    L.optix_ctx_destroy(ctx)
    assert np.isfinite(gpu.render_rgb(w, h, 1)).all()  # the lazily created renderer is unaffected


def test_render_edge_cases(gpu, oracle):
    """Tiny and odd image sizes, one sample, no samples, depth 1: same answers as the oracle's launch."""
    L = gpu.lib()
    L.pt_set_mode(gpu.MODE_STRICT)
    try:
        for pipeline in (0, 1):
            L.pt_set_pipeline(pipeline)
            for (w, h) in [(1, 1), (7, 3), (2, 65), (130, 1)]:
                L.pt_set_frame(0)
                img = gpu.render_rgb(w, h, 4)
                ref = oracle.render(w, h, 4, frame=1, mode=oracle.STRICT)[0]
                assert img.shape == (h, w, 3) and np.isfinite(img).all()
                close = (np.abs(img - ref) <= 2e-3 + 2e-3 * np.abs(ref)).all(axis=2).mean()
                assert close >= 0.85, (w, h, close)
            L.pt_set_frame(0)
            one = gpu.render_rgb(16, 12, 1)
            ref1 = oracle.render(16, 12, 1, frame=1, mode=oracle.STRICT)[0]
            assert (np.abs(one - ref1) <= 2e-3 + 2e-3 * np.abs(ref1)).all(axis=2).mean() > 0.9
            assert not gpu.render_rgb(16, 12, 0).any()  # no samples: zeros, no division by zero
            L.pt_set_frame(0)
            L.pt_reset_ray_counts()
            d1 = gpu.render_rgb(32, 24, 4, max_depth=1)
            rr, rs = gpu.ray_counts()
            o1, _, _, orays = oracle.render(32, 24, 4, max_depth=1, frame=1, mode=oracle.STRICT)
            assert rr == 32 * 24 * 4 == orays[0]  # depth 1: the camera ray only (plus its shadow ray)
            assert (np.abs(d1 - o1) <= 2e-3 + 2e-3 * np.abs(o1)).all(axis=2).mean() > 0.95
        assert L.optix_render_rgb(8, 8, 1, 4, C.c_float(0.0), 32, None) == -1  # NULL output (optix_wrapper.cpp:980)
    finally:
        L.pt_set_mode(gpu.MODE_CONVERGE)
        L.pt_set_pipeline(1)


# ------------------------------------------------------------------ adaptive sampling
@pytest.mark.parametrize("pipeline", [0, 1])
def test_adaptive_sampling_matches_oracle(gpu, oracle, pipeline):
    """noise_threshold > 0: same round cadence and noise test as the oracle's scheduling-independent
    restatement of optix_device.cu:1011-1060; per-pixel sample counts agree except where a noise value
    sits on the threshold, and the image stays within the tolerance of a full-spp render."""
    L = gpu.lib()
    w, h, spp, thr, mn, rnd = 96, 32, 256, 1.5, 32, 32
    L.pt_set_pipeline(pipeline)
    L.pt_set_mode(gpu.MODE_CONVERGE)
    L.pt_set_adaptive_round(rnd)
    L.pt_set_adaptive_chunks(1)  # one random stream per pixel, as in the oracle
    L.pt_set_frame(0)
    L.pt_reset_ray_counts()
    img = gpu.render_rgb(w, h, spp, noise_threshold=thr, min_adaptive_samples=mn)
    counts, rounds = gpu.sample_counts(w, h)
    rr, rs = gpu.ray_counts()
    o_img, _, o_counts, o_rays = oracle.render(w, h, spp, frame=1, noise_threshold=thr, min_adaptive=mn,
                                               mode=oracle.CONVERGE, adaptive_deterministic=1, adaptive_round=rnd)
    assert counts.min() >= mn and counts.max() <= spp and (counts % rnd == 0).all()
    assert 1 < rounds <= spp // rnd
    assert len(np.unique(counts)) > 2  # it adapts: neither all-min nor all-max
    # black side bars stop at the first test (their noise is 0)
    assert (counts[:, 0] == mn).all() and (o_counts[:, 0] == mn).all()
    same = (counts == o_counts).mean()
    assert same > 0.95, same  # measured 0.979 (0.986 with doubling rounds)
    assert abs(counts.mean() - o_counts.mean()) / o_counts.mean() < 0.05
    assert abs((rr + rs) - sum(o_rays)) / sum(o_rays) < 0.05
    assert abs(img.mean() - o_img.mean()) / o_img.mean() < 0.05  # early stops make the mean firefly-sensitive
    close = (np.abs(img - o_img) <= 2e-3 + 2e-2 * np.abs(o_img)).all(axis=2).mean()
    assert close > 0.7, close  # measured 0.75: the pixels whose counts differ, and single bright samples
    # doubling rounds (the default): 32, 64, 128, 256
    L.pt_set_adaptive_round(-32)
    L.pt_set_frame(0)
    gpu.render_rgb(w, h, spp, noise_threshold=thr, min_adaptive_samples=mn)
    c2, r2 = gpu.sample_counts(w, h)
    o2 = oracle.render(w, h, spp, frame=1, noise_threshold=thr, min_adaptive=mn, mode=oracle.CONVERGE,
                       adaptive_deterministic=1, adaptive_round=-32)[2]
    assert set(np.unique(c2)) <= {32, 64, 128, 256} and r2 <= 4
    assert (c2 == o2).mean() > 0.95
    L.pt_set_adaptive_round(rnd)
    # threshold so high that every pixel stops at the first test; so low that none stops
    L.pt_set_frame(0)
    gpu.render_rgb(w, h, spp, noise_threshold=1e9, min_adaptive_samples=mn)
    c_hi, r_hi = gpu.sample_counts(w, h)
    assert (c_hi == mn).all() and r_hi == 1
    L.pt_set_frame(0)
    full = gpu.render_rgb(w, h, 48, noise_threshold=1e-9, min_adaptive_samples=mn)
    c_lo, _ = gpu.sample_counts(w, h)
    assert (c_lo[:, w // 2] == 48).all() and (c_lo[:, 0] == mn).all()
    L.pt_set_frame(0)
    plain = gpu.render_rgb(w, h, 48)  # below 64 spp a pixel is one slot, as in an adaptive render
    lit = c_lo == 48  # pixels that never stopped took exactly the plain render's samples
    assert np.allclose(full[lit], plain[lit], rtol=1e-5, atol=1e-6)
    c_plain, r_plain = gpu.sample_counts(w, h)
    assert (c_plain == 48).all() and r_plain == 0
    L.pt_set_adaptive_chunks(0)


def test_adaptive_chunks(gpu):
    """By default an adaptive render splits a pixel's samples into chunks rendered side by side (more paths
    in flight, shorter tail per round).  Same rounds, same criterion: counts land on the round boundaries, the
    count map and the image agree with the one-stream render as two renders with different random numbers
    do, and the result is reproducible."""
    L = gpu.lib()
    w, h, spp, thr, mn = 96, 32, 256, 1.5, 32
    L.pt_set_mode(gpu.MODE_CONVERGE)
    L.pt_set_adaptive_round(-32)
    out = {}
    for chunks in (1, 0, 0, 4):
        L.pt_set_adaptive_chunks(chunks)
        L.pt_set_frame(0)
        img = gpu.render_rgb(w, h, spp, noise_threshold=thr, min_adaptive_samples=mn)
        counts, rounds = gpu.sample_counts(w, h)
        assert set(np.unique(counts)) <= {32, 64, 128, 256} and rounds <= 4
        assert (counts[:, 0] == mn).all()
        if chunks == 0 and 0 in out:
            assert np.array_equal(out[0][0], img) and np.array_equal(out[0][1], counts)
        out[chunks] = (img, counts)
    L.pt_set_adaptive_chunks(0)
    one, auto, four = out[1], out[0], out[4]
    assert not np.array_equal(one[0], auto[0])  # other random streams
    for other in (auto, four):
        assert (other[1] == one[1]).mean() > 0.6
        assert abs(other[1].mean() - one[1].mean()) / one[1].mean() < 0.06
        assert abs(other[0].mean() - one[0].mean()) / one[0].mean() < 0.05
    # fixed rounds of 10 samples: 2 chunks of 5 would leave fewer than 4 per chunk and round -> one chunk... 40: 8 chunks
    L.pt_set_adaptive_round(40)
    L.pt_set_frame(0)
    gpu.render_rgb(w, h, 200, noise_threshold=thr, min_adaptive_samples=mn)
    c40, _ = gpu.sample_counts(w, h)
    assert set(np.unique(c40)) <= {40, 80, 120, 160, 200} and (c40[:, 0] == 40).all()
    L.pt_set_adaptive_round(-32)


def test_adaptive_shares_accumulate(gpu):
    """Sample-sharded adaptive render: two shares of 128 samples, each stopping pixels on its own noise
    estimate, add up to sums and counts whose ratio is the image."""
    import torch
    from ptamd import sharded
    L = gpu.lib()
    w, h, per, thr = 96, 32, 128, 1.5
    dev = torch.device("cuda:0")

    def share(sums, counts, begin, count):
        torch.cuda.synchronize()
        gpu.render_adaptive_accumulate_device(sums, counts, w, h, begin, count, 32, 5, thr, 32)

    sums = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
    counts = torch.zeros(w * h, dtype=torch.float32, device=dev)
    share(sums, counts, 0, per)
    c0 = counts.clone()
    share(sums, counts, per, per)
    c = counts.cpu().numpy().reshape(h, w)
    c1 = c - c0.cpu().numpy().reshape(h, w)
    assert (c[:, 0] == 64).all()  # black bars: both shares stop at their first test
    assert c.max() == 2 * per and ((c0.cpu().numpy() % 32) == 0).all() and ((c1 % 32) == 0).all()
    assert 0.6 < (c1 == c0.cpu().numpy().reshape(h, w)).mean() < 1.0  # same criterion, different samples
    img = (sums.cpu().numpy().reshape(h, w, 3) / c[..., None])
    L.pt_set_frame(4)
    full = gpu.render_rgb(w, h, 2 * per)  # frame 5, fixed count
    assert abs(img.mean() - full.mean()) / full.mean() < 0.05
    # the helper (world size 1) gives the same image from one share
    mean, cnt = sharded.render_sharded_adaptive(share, w, h, 0, per, device=dev)
    assert torch.equal(cnt.flatten(), c0.flatten())
    # threshold 0: fixed count, same sums as the plain accumulate entry point
    s2 = torch.zeros_like(sums); n2 = torch.zeros_like(counts); s3 = torch.zeros_like(sums)
    torch.cuda.synchronize()
    gpu.render_adaptive_accumulate_device(s2, n2, w, h, 0, 48, 32, 5, 0.0, 32)
    gpu.render_accumulate_device(s3, w, h, 0, 48, 32, 5)
    assert torch.equal(s2, s3) and (n2 == 48).all()


def test_adaptive_shared_one_rank_is_plain_adaptive(gpu):
    """The shared-statistics entry point with one rank (no reduce) makes the decisions of the per-share one:
    same sums, same counts, bit for bit; the statistics it returns are those of the render."""
    import torch
    from ptamd import sharded
    w, h, spp, thr = 96, 32, 128, 1.5
    dev = torch.device("cuda:0")
    gpu.lib().pt_set_adaptive_round(-32)  # doubling rounds, the library's default
    sums = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
    counts = torch.zeros(w * h, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu.render_adaptive_accumulate_device(sums, counts, w, h, 0, spp, 32, 5, thr, 32)
    calls = []
    stats = torch.zeros(w * h * 8, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu.render_adaptive_shared_device(stats, w, h, 0, spp, spp, 32, 5, thr, 32, reduce=lambda: calls.append(1))
    v = stats.view(w * h, 8)
    assert torch.equal(v[:, 0:3].reshape(-1), sums) and torch.equal(v[:, 3], counts)
    assert (v[:, 7] == 0).all() and (v[:, 4:7] >= 0).all()
    c = counts.cpu().numpy()
    assert set(np.unique(c)) <= {32.0, 64.0, 128.0} and len(np.unique(c)) > 1
    assert len(calls) == 3  # after 32 and 64 samples, and the totals at the end
    # a longer render, whose later rounds give the last chunk its smaller part: still the same bits from both
    big, big_n = torch.zeros_like(sums), torch.zeros_like(counts)
    torch.cuda.synchronize()
    gpu.render_adaptive_accumulate_device(big, big_n, w, h, 0, 512, 32, 5, thr, 32)
    torch.cuda.synchronize()
    gpu.render_adaptive_shared_device(stats, w, h, 0, 512, 512, 32, 5, thr, 32, reduce=None)
    assert torch.equal(v[:, 0:3].reshape(-1), big) and torch.equal(v[:, 3], big_n) and float(big_n.max()) == 512.0
    # NULL reduce and the helper give the same
    renderer = sharded.DeviceSharedAdaptiveRenderer(w, h, 32, 5, thr, 32, dev)
    mean, cnt = sharded.render_sharded_adaptive_shared(renderer, w, h, 0, spp, spp, device=dev)
    assert torch.equal(cnt.reshape(-1), counts)
    assert torch.equal(mean.reshape(-1, 3), sums.view(-1, 3) / counts.unsqueeze(1))
    # argument checks; an exception in reduce comes back as itself
    L = gpu.lib()
    assert L.pt_render_adaptive_shared_device(w, h, 0, spp, spp, 32, 5, 1.0, 32, None, gpu.REDUCE_FN(), None) == -1
    assert L.pt_render_adaptive_shared_device(w, h, 0, spp, spp - 1, 32, 5, 1.0, 32, stats.data_ptr(), gpu.REDUCE_FN(), None) == -1
    assert L.pt_render_adaptive_shared_device(w, h, 0, spp, spp, 32, 5, 0.0, 32, stats.data_ptr(), gpu.REDUCE_FN(), None) == -1

    def boom():
        raise KeyError("reduce failed")
    with pytest.raises(KeyError):
        gpu.render_adaptive_shared_device(stats, w, h, 0, spp, spp, 32, 5, thr, 32, reduce=boom)
    out = gpu.render_rgb(w, h, 8)  # the renderer is usable afterwards
    assert np.isfinite(out).all()


def test_stopped_pixels_satisfy_the_noise_criterion(gpu):
    """The stopping rule of optix_device.cu:1011-1049 recomputed outside the library: a pixel that stopped keeps the
    statistics it stopped with, so its noise -- relative standard deviation of luminance, or of its own Bayer channel
    if that is larger, from (sum, sum of squares, n) in float64 -- must lie below the threshold, and so must that of
    every neighbour that stopped with it or before; pixels that never reached the minimum count never stop."""
    import torch
    w, h, spp, thr, mn = 96, 54, 256, 0.6, 32
    dev = torch.device("cuda:0")
    L = gpu.lib()
    L.pt_set_adaptive_round(-32)
    stats = torch.zeros(w * h * 8, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu.render_adaptive_shared_device(stats, w, h, 0, spp, spp, 32, 9, thr, mn, reduce=None)
    v = stats.view(h, w, 8).cpu().numpy().astype(np.float64)
    s1, n, s2 = v[..., 0:3], v[..., 3], v[..., 4:7]
    assert set(np.unique(n)) <= {32.0, 64.0, 128.0, 256.0} and n.min() >= mn
    mean = s1 / n[..., None]
    var = np.maximum(s2 - s1 * mean, 0.0) / (n[..., None] - 1.0)
    lw = np.array([0.2126, 0.7152, 0.0722])
    noise = np.sqrt(np.maximum(var @ lw, 0.0)) / np.maximum(mean @ lw, 1e-6)
    yy, xx = np.mgrid[0:h, 0:w]
    ch = np.where(yy % 2 == 0, np.where(xx % 2 == 0, 0, 1), np.where(xx % 2 == 0, 1, 2))  # pattern 0 (optix_device.cu:902-912)
    mean_b = np.take_along_axis(mean, ch[..., None], 2)[..., 0]
    var_b = np.take_along_axis(var, ch[..., None], 2)[..., 0]
    noise = np.maximum(noise, np.sqrt(var_b) / np.maximum(mean_b, 1e-6))
    stopped = n < spp
    assert 0.2 < stopped.mean() < 0.98  # the render has both kinds of pixel
    slack = 1e-4 * thr  # float32 statistics recomputed in float64
    assert (noise[stopped] < thr + slack).all(), noise[stopped].max()
    # neighbours that stopped no later than the pixel did held their final statistics when it was tested
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            nb_noise = np.roll(noise, (dy, dx), (0, 1))
            nb_n = np.roll(n, (dy, dx), (0, 1))
            inside = (yy - dy >= 0) & (yy - dy < h) & (xx - dx >= 0) & (xx - dx < w)
            sel = stopped & inside & (nb_n <= n)
            assert (nb_noise[sel] < thr + slack).all()
    # and a pixel all of whose neighbourhood stayed noisy did not stop: where even the own noise is above the threshold
    assert not (stopped & (noise >= thr + slack)).any()


_SHARED_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import ptamd
from ptamd import sharded
rank, world = int(sys.argv[3]), 2
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=rank, world_size=world)
w, h, per, thr = 96, 32, 64, 1.5
dev = torch.device("cuda:0")
renderer = sharded.DeviceSharedAdaptiveRenderer(w, h, 32, 5, thr, 32, dev)
begin, count = sharded.chunk_of(rank, per)
mean, cnt = sharded.render_sharded_adaptive_shared(renderer, w, h, begin, count, world * per, device=dev)
own, rounds = ptamd.sample_counts(w, h)
np.savez(sys.argv[4] + "/rank%d.npz" % rank, mean=mean.cpu().numpy(), cnt=cnt.cpu().numpy(), own=own, rounds=rounds)
dist.barrier()
dist.destroy_process_group()
'''


def test_adaptive_shared_two_ranks(gpu, tmp_path):
    """Two ranks (two processes on this GPU, gloo) render 64 samples each of a 128-sample adaptive render with
    the noise test on their summed statistics: both end with the same image and the same counts, every
    pixel took the same number of samples on both ranks, and rounds fall at 32, 64 and 128 samples of both
    ranks together."""
    import socket
    import sys
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    script = tmp_path / "shared_worker.py"
    script.write_text(_SHARED_WORKER)
    pkg = os.path.join(ROOT, "rust-pathtracer_amd")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), pkg, str(port), str(r), str(tmp_path)], env=env)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    a, b = (np.load(tmp_path / f"rank{r}.npz") for r in range(2))
    assert np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["cnt"], b["cnt"])
    assert np.array_equal(a["own"], b["own"]) and int(a["rounds"]) == int(b["rounds"]) == 3
    assert np.array_equal(a["own"] * 2, a["cnt"].astype(np.int32))
    cnt = a["cnt"]
    assert set(np.unique(cnt)) <= {32.0, 64.0, 128.0} and len(np.unique(cnt)) > 1
    assert (cnt[:, 0] == 32).all()  # black bars stop at the first shared test: 16 samples on each rank
    gpu.lib().pt_set_adaptive_round(-32)  # what the fresh worker processes ran with
    gpu.lib().pt_set_frame(4)
    full = gpu.render_rgb(96, 32, 128)  # frame 5, fixed count
    assert abs(a["mean"].mean() - full.mean()) / full.mean() < 0.05
    # one rank with the same total makes its decisions at the same totals: similar count map
    import torch
    stats = torch.zeros(96 * 32 * 8, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    gpu.render_adaptive_shared_device(stats, 96, 32, 0, 128, 128, 32, 5, 1.5, 32)
    one = stats.view(32, 96, 8)[..., 3].cpu().numpy()
    assert (one == cnt).mean() > 0.6


# ------------------------------------------------------------------ a16 / a17 demosaic + EXR
@pytest.mark.parametrize("algo,name", [(0, "amaze"), (1, "bilinear")])
def test_demosaic_bit_exact(gpu, oracle, algo, name):
    rng = np.random.default_rng(algo)
    # (from 60 x 124 on an image has tiles that lie inside it and take the kernel's other routine)
    for shape in [(1, 1), (1, 7), (2, 2), (5, 3), (48, 64), (135, 241), (60, 124), (59, 300), (301, 517)]:
        for pattern in range(4):
            raw = rng.uniform(0, 4, shape).astype(np.float32)
            raw[rng.uniform(size=shape) < 0.05] = 0.0
            got = gpu.demosaic(raw, pattern, algo)
            want = oracle.demosaic(raw, pattern, name)
            assert np.array_equal(got, want), (shape, pattern)


def test_demosaic_random_shapes_and_unaligned_buffers(gpu, oracle):
    """Random image sizes (border tiles only, one interior tile, many; odd and even widths) through the host-buffer
    entry point, and device buffers that start 4 bytes off an 8-byte boundary through pt_demosaic_device (the
    interior routine then stores single words instead of pairs): all bit-identical to the oracle's loops."""
    import torch
    L = gpu.lib()
    rng = np.random.default_rng(77)
    shapes = [(int(rng.integers(1, 200)), int(rng.integers(1, 420))) for _ in range(24)]
    shapes += [(60, 123), (60, 124), (59, 124), (88, 184), (87, 183), (116, 244)]
    for k, (h, w) in enumerate(shapes):
        pattern = k % 4
        raw = rng.uniform(0, 2, (h, w)).astype(np.float32)
        raw[rng.uniform(size=(h, w)) < 0.1] = 0.0
        want = oracle.demosaic(raw, pattern, "amaze")
        assert np.array_equal(gpu.demosaic(raw, pattern, 0), want), (h, w, pattern)
        if k % 3 == 0:
            d_raw = torch.zeros(h * w + 1, dtype=torch.float32, device="cuda:0")
            d_raw[1:] = torch.from_numpy(raw.ravel()).to("cuda:0")
            d_out = torch.zeros(h * w * 3 + 1, dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            assert L.pt_demosaic_device(d_raw.data_ptr() + 4, w, h, pattern, 0, d_out.data_ptr() + 4) == 0
            torch.cuda.synchronize()
            assert float(d_out[0]) == 0.0
            assert np.array_equal(d_out[1:].cpu().numpy().reshape(h, w, 3), want), (h, w, pattern, "offset")


def test_demosaic_full_size_properties(gpu, oracle):
    """At 3840x2160 (the Bayer configuration's size) the oracle is only run on a crop; the whole
    image is checked through properties: green kept at green sites, non-negative, flat interior."""
    rng = np.random.default_rng(8)
    h, w = 2160, 3840
    raw = rng.uniform(0.1, 2.0, (h, w)).astype(np.float32)
    out = gpu.demosaic(raw, 0, 0)
    assert out.shape == (h, w, 3) and out.min() >= 0 and np.isfinite(out).all()
    assert np.array_equal(out[0::2, 1::2, 1], raw[0::2, 1::2]) and np.array_equal(out[1::2, 0::2, 1], raw[1::2, 0::2])
    crop = oracle.demosaic(raw[:64, :64], 0, "amaze")
    assert np.array_equal(out[:60, :60], crop[:60, :60])  # away from the crop's artificial border
    flat = gpu.demosaic(np.full((h, w), 0.5, np.float32), 2, 0)
    assert np.allclose(flat[3:-3, 3:-3], 0.5, atol=1e-5)


def test_bayer_render_and_demosaic_chain_at_full_size(gpu):
    """The Bayer configuration as the host runs it (src/main.rs:512-555), at its own size: optix_render_bayer_f32 at
    3840 x 2160, pattern 0, then the gradient demosaic of that mosaic on the device.  No CPU render of this size
    exists, so the chain is checked through what must hold at any size: the mosaic is, bit for bit, the CFA channel
    of the RGB render of the same frame; the side bars (open front) are zero in the mosaic and stay zero through the
    demosaic; green sites keep their raw value; and the 40 x 40 block means of mosaic and demosaic agree with the
    oracle's converged 96 x 54 image of the same view (same aspect, so a block is a pixel of it)."""
    import torch
    L = gpu.lib()
    w, h, spp = 3840, 2160, 256  # (the first sample of a pixel's stream is ~9 % bright, D12: 2 % of a 16-spp mean)
    L.pt_set_frame(1)  # -> frame 2, the frame of the reference's Bayer call
    raw = gpu.render_bayer(w, h, spp, pattern=0)
    L.pt_set_frame(1)
    rgb = gpu.render_rgb(w, h, spp)
    assert raw.shape == (h, w) and np.isfinite(raw).all() and raw.min() >= 0
    ys, xs = np.mgrid[0:h, 0:w]
    ch = (xs & 1) + (ys & 1)  # RGGB: 0 / 1 / 1 / 2
    assert np.array_equal(raw, np.maximum(np.take_along_axis(rgb, ch[..., None], 2)[..., 0], 0))
    assert raw[:, :500].max() == 0 and raw[:, -500:].max() == 0
    # demosaic on the device (pt_demosaic_device), as a host that keeps the mosaic resident would call it
    d_raw = torch.from_numpy(raw).to("cuda:0").contiguous()
    d_out = torch.empty(h * w * 3, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    assert L.pt_demosaic_device(d_raw.data_ptr(), w, h, 0, 0, d_out.data_ptr()) == 0
    torch.cuda.synchronize()
    dem = d_out.cpu().numpy().reshape(h, w, 3)
    assert np.array_equal(dem, gpu.demosaic(raw, 0, 0))  # the host-buffer entry point runs the same kernel
    assert np.isfinite(dem).all() and dem.min() >= 0
    assert np.array_equal(dem[0::2, 1::2, 1], raw[0::2, 1::2]) and np.array_equal(dem[1::2, 0::2, 1], raw[1::2, 0::2])
    assert dem[:, :496].max() == 0 and dem[:, -496:].max() == 0
    # against the oracle's converged image: the 4K frame reduced 40 times with the pixel filter is the 96 x 54 golden
    # (its outermost pixels aside, where the reference clamps the jittered position to the image)
    ref = np.load(os.path.join(GOLD, "converge_96x54_spp4096_frame1.npz"))["rgb"]
    dem_b, rgb_b = tent_downsample(dem, 40), tent_downsample(rgb, 40)
    inner = (slice(1, -1), slice(1, -1))
    lit = ref[inner].sum(axis=2) > 0.05
    # (what is left at 256 spp over 1600 pixels is the 4096-spp golden's own noise, 0.029 per pixel)
    for img_b, tol_med, tol_rmse in ((rgb_b, 0.03, 0.05), (dem_b, 0.03, 0.055)):
        rel = np.abs(img_b[inner] - ref[inner])[lit] / (ref[inner][lit] + 0.01)
        assert np.median(rel) < tol_med, np.median(rel)
        assert rel_rmse(img_b[inner], ref[inner]) < tol_rmse, rel_rmse(img_b[inner], ref[inner])
    for c in range(3):
        assert abs(dem[..., c].mean() - ref[..., c].mean()) / ref[..., c].mean() < 0.012
        assert abs(rgb[..., c].mean() - ref[..., c].mean()) / ref[..., c].mean() < 0.004
    # the mosaic's own channel planes (quarter resolution each) against the golden's channels
    for (dy, dx, c) in ((0, 0, 0), (0, 1, 1), (1, 0, 1), (1, 1, 2)):
        plane = raw[dy::2, dx::2].reshape(54, 20, 96, 20).mean(axis=(1, 3))
        assert abs(plane.mean() - ref[..., c].mean()) / ref[..., c].mean() < 0.005


def test_rays_that_start_on_a_plane_keep_their_small_known_bias(gpu):
    """A path that leaves the glass box through its bottom face restarts exactly on the floor plane (the reference's
    1e-3 offset, optix_device.cu:469-475): whether it sees the floor is rounding noise in the oracle and decided by
    the fast triangle test's own sign here, so such paths always pass through the floor where about half of the
    oracle's bounce.  The size of that difference is pinned here so that it cannot grow unnoticed: the strict-mode
    launch traces between 97.5 % and 100 % of the oracle's rays (radiance and shadow), and the converged image keeps
    at least 99.9 % of the oracle's mean in every channel (measured 0.9994-0.9999 at 1024 spp; two independent
    frames are averaged to keep noise out of the ratio)."""
    L = gpu.lib()
    g = np.load(os.path.join(GOLD, "strict_32x32_spp4_frame1.npz"))
    for pipeline in (0, 1):
        L.pt_set_pipeline(pipeline)
        L.pt_set_mode(gpu.MODE_STRICT)
        L.pt_set_frame(0)
        L.pt_reset_ray_counts()
        gpu.render_rgb(32, 32, 4)
        rr, rs = gpu.ray_counts()
        assert 0.975 <= rr / g["rays"][0] <= 1.0, (rr, g["rays"][0])
        assert 0.975 <= rs / g["rays"][1] <= 1.0, (rs, g["rays"][1])
    L.pt_set_mode(gpu.MODE_CONVERGE)
    L.pt_set_pipeline(1)
    ref = np.load(os.path.join(GOLD, "converge_96x54_spp4096_frame1.npz"))["rgb"].astype(np.float64)
    acc = np.zeros_like(ref)
    frames = (10, 20, 30, 40)
    for f in frames:
        L.pt_set_frame(f)
        acc += gpu.render_rgb(96, 54, 1024)
    ratio = (acc / len(frames)).mean(axis=(0, 1)) / ref.mean(axis=(0, 1))
    assert (ratio >= 0.999).all() and (ratio <= 1.0015).all(), ratio


def test_driver_writes_exr_files(gpu, tmp_path):
    """The C++ host driver (the restatement of src/main.rs) end to end at a small size."""
    exe = os.path.join(gpu.ROOT, "pt_main")
    a, b = str(tmp_path / "pt.exr"), str(tmp_path / "pt_bayer.exr")
    out = subprocess.check_output([exe, "--width=64", "--height=48", "--spp=16", "--noise-threshold=0",
                                   f"--out={a}", f"--out-bayer={b}"], text=True, timeout=120)
    assert "Saved" in out and "Demosaic (AMaZE)" in out and "Exposure multiplier=1.000000" in out
    import exr_reader  # a parser of the file layout that shares nothing with the writer
    rgb, dem = exr_reader.read_rgb(a)[0], exr_reader.read_rgb(b)[0]
    assert np.array_equal(rgb, gpu.read_exr(a)) and np.array_equal(dem, gpu.read_exr(b))
    assert rgb.shape == (48, 64, 3) and dem.shape == (48, 64, 3)
    # pt.exr holds the render itself: the driver's first render call is frame 1 of a fresh renderer
    L = gpu.lib()
    L.pt_set_frame(0)
    assert np.array_equal(rgb, gpu.render_rgb(64, 48, 16))
    assert rgb[:, :1].max() == 0 and rgb.mean() > 0.05 and dem.mean() > 0.05
    assert abs(dem.mean() - rgb.mean()) / rgb.mean() < 0.2
    # the reference's defaults (adaptive sampling on) with the guiding loop: train, update, snapshot, render
    # (src/guiding.rs:222-270 through the stand-in trainer), bilinear demosaic
    a2, b2 = str(tmp_path / "g.exr"), str(tmp_path / "g_bayer.exr")
    out = subprocess.check_output([exe, "--width=64", "--height=48", "--spp=64", "--guiding", "--guide-grid=2x2x2",
                                   "--demosaic=bilinear", f"--out={a2}", f"--out-bayer={b2}"], text=True, timeout=120)
    assert "Saved" in out
    g_rgb, g_dem = gpu.read_exr(a2), gpu.read_exr(b2)
    assert np.isfinite(g_rgb).all() and np.isfinite(g_dem).all() and g_rgb.mean() > 0.02 and g_dem.mean() > 0.02


def test_layout_matches_reference_render_at_its_own_size(gpu):
    """The reference's committed pt.png (1920x1440, region means in tests/golden/ptpng_regions.json) against
    a GPU render of the same size: black side bars, the coloured walls on the same sides with the same
    dominant channel, a neutral back wall, the light brighter than everything else."""
    import json
    meta = json.load(open(os.path.join(GOLD, "ptpng_regions.json")))
    reg = meta["regions"]
    w, h = meta["size"]
    L = gpu.lib()
    L.pt_set_frame(0)
    rgb = gpu.render_rgb(w, h, 64)

    def mean(name):
        x0, x1, y0, y1 = reg[name]["frac"]
        return rgb[int(y0 * h):int(y1 * h), int(x0 * w):int(x1 * w)].mean(axis=(0, 1))

    for name in ("left_bar", "right_bar"):
        assert reg[name]["mean_rgb8"] == [0.0, 0.0, 0.0] and np.all(mean(name) == 0.0)
    lw, rw, bw, light = mean("left_wall"), mean("right_wall"), mean("back_wall"), mean("light")
    assert np.argmax(lw) == int(np.argmax(reg["left_wall"]["mean_rgb8"]))
    assert np.argmax(rw) == int(np.argmax(reg["right_wall"]["mean_rgb8"]))
    assert np.argmin(lw) == int(np.argmin(reg["left_wall"]["mean_rgb8"]))
    assert np.argmin(rw) == int(np.argmin(reg["right_wall"]["mean_rgb8"]))
    assert bw.max() / bw.min() < 1.3 < min(lw.max() / lw.min(), rw.max() / rw.min())
    assert light.min() > 3 * bw.max()
    # brightness order of the regions is the one of the reference image
    order_ref = sorted(("left_wall", "right_wall", "back_wall", "light"), key=lambda k: sum(reg[k]["mean_rgb8"]))
    order_gpu = sorted(("left_wall", "right_wall", "back_wall", "light"), key=lambda k: float(mean(k).sum()))
    assert order_ref[-1] == order_gpu[-1] == "light"


# ------------------------------------------------------------------ full-size properties
def test_large_scene_at_its_own_size(gpu):
    """The 12.8 M-triangle synthetic scene (40 000 spheres of 320 triangles in the room; BVH and records 1.1 GB in
    HBM) at 3840 x 2160: size-independent properties, since no CPU render of this exists.  Closed box: no ray
    heading away from the open front escapes; renders are finite, deterministic, the same bits from the staged,
    the fused and the mixed pipeline; and the image mean is that of the same 40 000 objects tessellated 16 times
    coarser (same seed, same places and materials) within a few per cent."""
    L = gpu.lib()
    w, h = 3840, 2160
    try:
        L.pt_set_frame(11)
        box_before = gpu.render_rgb(w, h, 4)
        assert L.pt_last_render_chunks() == 1
        L.pt_scene_synthetic(42, 40000, 2)
        info = gpu.scene_info()
        assert info["triangles"] == 12 + 40000 * 320 and info["bvh_bytes"] > 1_000_000_000 and info["depth"] >= 9
        o, d = _random_rays(2_000_000, 21)
        prim, t, b1, b2 = gpu.trace_rays(o, d)
        away = d[:, 1] < 0
        assert (prim[away] >= 0).all() and np.isfinite(t[prim >= 0]).all()
        assert (prim >= 12).mean() > 0.5  # most rays end on a sphere, not on a wall
        L.pt_set_defer_shadow(0)
        imgs = []
        for pipeline in (-1, 0, 1):
            L.pt_set_pipeline(pipeline)
            L.pt_set_frame(11)
            L.pt_reset_ray_counts()
            imgs.append(gpu.render_rgb(w, h, 2))
            assert L.pt_last_render_chunks() == 2  # a large scene keeps every sample in flight, whatever the pipeline
            rays = gpu.ray_counts()
        assert np.isfinite(imgs[0]).all() and imgs[0].min() >= 0 and imgs[0].mean() > 0.01
        assert np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[1], imgs[2])
        assert 2.0 < rays[0] / (w * h * 2) < 12.0
        L.pt_set_pipeline(-1)
        L.pt_set_frame(11)
        again = gpu.render_rgb(w, h, 2)
        L.pt_set_frame(11)
        assert np.array_equal(again, gpu.render_rgb(w, h, 2))  # deterministic
        fine_mean = again.mean(axis=(0, 1))
        L.pt_scene_synthetic(42, 40000, 0)
        L.pt_set_frame(11)
        coarse_mean = gpu.render_rgb(w, h, 2).mean(axis=(0, 1))
        assert (np.abs(fine_mean - coarse_mean) / coarse_mean < 0.06).all(), (fine_mean, coarse_mean)
        # the path state that grew for the large scene changes nothing for a small one
        L.pt_scene_cornell()
        L.pt_set_frame(11)
        assert np.array_equal(box_before, gpu.render_rgb(w, h, 4)) and L.pt_last_render_chunks() == 1
    finally:
        L.pt_scene_cornell()
        L.pt_set_pipeline(1)
        L.pt_set_defer_shadow(DEFER_DEFAULT)


def test_adaptive_render_of_a_scene_that_runs_staged(gpu):
    """Adaptive sampling on a scene beyond the fused kernel's size (160 000 triangles, 13.6 MiB of tree): the render
    takes 32 slots per pixel (one sample per slot in a round of 32) whatever was rendered before it, gives the same
    bits from the staged and the fused kernels and when repeated after other renders, adapts, and stays close to the
    fixed-count render of the same scene."""
    L = gpu.lib()
    w, h, spp, thr, mn = 320, 180, 256, 0.2, 32
    try:
        L.pt_scene_synthetic(42, 500, 2)
        assert gpu.scene_info()["bvh_bytes"] > 8 << 20
        L.pt_set_defer_shadow(0)
        imgs, cnts = [], []
        for pipeline in (0, 1, -1):
            L.pt_set_pipeline(pipeline)
            L.pt_set_frame(5)
            imgs.append(gpu.render_rgb(w, h, spp, noise_threshold=thr, min_adaptive_samples=mn))
            assert L.pt_last_render_chunks() == 32
            cnts.append(gpu.sample_counts(w, h)[0])
            if pipeline == 0:  # something else in between: a long fixed-count render that grows the state further
                L.pt_set_frame(9)
                fixed = gpu.render_rgb(w, h, spp)
                assert L.pt_last_render_chunks() == 64
        assert np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[0], imgs[2])
        assert np.array_equal(cnts[0], cnts[1]) and np.array_equal(cnts[0], cnts[2])
        c = cnts[0]
        assert c.min() >= mn and c.max() <= spp and (c % 32 == 0).all() and len(np.unique(c)) > 2
        assert np.isfinite(imgs[0]).all()
        assert abs(imgs[0].mean() - fixed.mean()) / fixed.mean() < 0.02
        blocks = lambda a: a.reshape(h // 20, 20, w // 20, 20, 3).mean(axis=(1, 3))
        assert np.abs(blocks(imgs[0]) - blocks(fixed)).max() / fixed.mean() < 0.15
    finally:
        L.pt_scene_cornell()
        L.pt_set_pipeline(1)
        L.pt_set_defer_shadow(DEFER_DEFAULT)


def test_one_sample_chunks_are_independent_between_pixels(gpu):
    """A scene that runs staged renders 64 spp as 64 chunks of one sample, each with a stream of its own.  The error of
    such a render must be noise that averages out over neighbouring pixels: the variance of 20 x 20 block means is
    1/400 of the per-pixel variance; and what a sample costs in rays does not depend on how the samples are cut into
    chunks.  (Streams that start from poorly mixed seeds give every pixel of the image the same first random number per
    sample: the second check fails by 2 % with chunk_seed's finaliser taken out, the first one does not notice.)"""
    L = gpu.lib()
    w, h, spp = 320, 180, 64
    try:
        L.pt_scene_synthetic(42, 500, 2)
        imgs = []
        for frame in (1, 2, 3):
            L.pt_set_frame(frame)
            imgs.append(gpu.render_rgb(w, h, spp).astype(np.float64))
            assert L.pt_last_render_chunks() == 64
        ratios = []
        for a, b in ((0, 1), (1, 2), (0, 2)):
            d = np.clip(imgs[a] - imgs[b], -1.0, 1.0)  # fireflies out of the second moments
            v_pix = (d ** 2).mean()
            blk = d.reshape(h // 20, 20, w // 20, 20, 3).mean(axis=(1, 3))
            ratios.append((blk ** 2).mean() * 400 / v_pix)
        assert all(0.6 < r < 1.6 for r in ratios), ratios
        # ... and the rays a sample takes on average do not depend on how the samples are cut into chunks: 16 chunks of
        # one sample against 64 chunks of 16 (0.07 % of noise at this size; 2 % apart with the shared first draws).
        # Sample ranges that start after sample 0: the stream that starts there is the reference's, whose first draw
        # is the same all over the frame (1 % on the average of 16 samples).
        import torch
        sums = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda:0")
        per_sample = []
        for begin, n in ((1024, 16), (2048, 1024)):
            L.pt_reset_ray_counts()
            gpu.render_accumulate_device(sums, w, h, begin, n, 32, 4)
            per_sample.append(sum(gpu.ray_counts()) / (w * h * n))
        assert abs(per_sample[0] / per_sample[1] - 1.0) < 4e-3, per_sample
    finally:
        L.pt_scene_cornell()


def test_instanced_scene_traces_and_renders_like_its_flat_copy(gpu, oracle):
    """A scene made of instances (two-level tree, rays taken to object space, normals brought back through the inverse
    transpose, a material per instance) against the same scene written out as world-space triangles: the closest
    hits are the oracle's brute-force hits, and the renders agree as two renders of one scene do."""
    L = gpu.lib()
    rng = np.random.default_rng(5)
    try:
        L.pt_scene_synthetic_instanced(42, 125, 3)
        assert L.pt_scene_instance_count() == 126
        o = rng.uniform(-0.95, 0.95, (20000, 3)).astype(np.float32)
        o[:, 2] = rng.uniform(0.02, 1.98, len(o))
        d = rng.normal(size=(len(o), 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        prim, t, b1, b2, inst = gpu.trace_rays_instances(o, d)
        tmax = rng.uniform(0.05, 2.0, len(o)).astype(np.float32)
        occ = gpu.trace_rays_instances(o, d, tmax=tmax, any_hit=True)[0]
        L.pt_set_frame(0)
        a = [gpu.render_rgb(96, 54, 256) for _ in range(2)]
        info = gpu.scene_info()
        assert info["triangles"] == 12 + 1280  # the meshes only
        assert L.pt_render_accumulate_device_noalloc(16, 16, 0, 4, 8, 1, None) != 0  # refused, not rendered wrongly
        # ... and its flat copy
        assert L.pt_scene_flatten() == 12 + 125 * 1280 and L.pt_scene_instance_count() == 0
        fprim, ft, fb1, fb2 = gpu.trace_rays(o, d)
        focc = gpu.trace_rays(o, d, tmax=tmax, any_hit=True)[0]
        flat = np.where(prim < 0, -1, np.where(inst == 0, prim, 12 + (inst - 1) * 1280 + (prim - 12)))
        assert (flat != fprim).mean() < 1e-3 and ((fprim >= 0) == (prim >= 0)).all()
        same = flat == fprim
        assert np.abs(t - ft)[same & (fprim >= 0)].max() < 3e-5 and np.abs(t - ft)[fprim >= 0].max() < 1e-4
        assert (occ != focc).mean() < 5e-4
        V, I, _, _ = gpu.scene_arrays()
        oracle.lib().orc_scene_set(P(V), len(V), P(I, C.c_uint32), len(I), None)
        fp = C.POINTER(C.c_float)
        sel = np.arange(0, len(o), 10)
        want = np.array([oracle.lib().orc_trace_closest(o[i].ctypes.data_as(fp), d[i].ctypes.data_as(fp), 0.0, 1e16).prim for i in sel])
        assert (flat[sel] != want).mean() < 2e-3
        L.pt_set_frame(0)
        b = [gpu.render_rgb(96, 54, 256) for _ in range(2)]
        L.pt_set_pipeline(0)
        L.pt_set_frame(0)
        c = gpu.render_rgb(96, 54, 256)
        assert np.array_equal(c, b[0]) or rel_rmse(c, b[0]) < 1e-4  # flat scene: staged = default pipeline
        for ch in range(3):
            ma, mb = [x[..., ch].mean() for x in a], [x[..., ch].mean() for x in b]
            sigma = max(abs(ma[0] - ma[1]), abs(mb[0] - mb[1])) / np.sqrt(2) + 2e-3 * mb[0]
            assert abs(np.mean(ma) - np.mean(mb)) < 3 * sigma, (ch, ma, mb)
        assert rel_rmse(a[0], b[0]) < 1.6 * rel_rmse(b[1], b[0]) + 0.01  # as far apart as two renders of the flat scene
    finally:
        oracle.lib().orc_scene_set(None, 0, None, 0, None)
        L.pt_set_pipeline(-1)
        L.pt_scene_cornell()


def test_full_size_1080p_properties(gpu):
    """The 1920x1080 headline configuration at reduced spp: properties that do not need a CPU render."""
    L = gpu.lib()
    L.pt_set_frame(0)
    L.pt_reset_ray_counts()
    w, h, spp = 1920, 1080, 32
    img = gpu.render_rgb(w, h, spp)
    assert np.isfinite(img).all() and img.min() >= 0
    assert img[:, :250].max() == 0 and img[:, -250:].max() == 0  # open front -> black side bars
    rr, rs = gpu.ray_counts()
    paths = w * h * spp
    assert 4.0 < rr / paths < 9.0 and 0.4 < rs / rr < 0.7  # oracle: ~6.3 radiance rays/path, ~0.54 shadow per radiance
    # the image mean agrees with the oracle's converged 96x54 image of the same view
    ref = np.load(os.path.join(GOLD, "converge_96x54_spp4096_frame1.npz"))["rgb"]
    assert np.all(np.abs(img.mean(axis=(0, 1)) - ref.mean(axis=(0, 1))) / ref.mean(axis=(0, 1)) < 0.02)
    small = tent_downsample(img, 20)  # the pixel filter of a 96 x 54 render of the same view
    assert rel_rmse(small[1:-1, 1:-1], ref[1:-1, 1:-1]) < 0.06
    L.pt_set_frame(0)
    assert np.array_equal(img, gpu.render_rgb(w, h, spp))  # deterministic
