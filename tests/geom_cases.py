"""Awkward geometry for the traversal tests (CPU host build and GPU): degenerate triangles, slivers and
millimetre triangles far from the origin, with a float64 ground truth that knows which answers the float32
rounding of the inputs leaves open."""
import numpy as np


def awkward_scene():
    rng = np.random.default_rng(11)
    verts, idx = [], []

    def add(a, b, c):
        base = len(verts)
        verts.extend([a, b, c])
        idx.append([base, base + 1, base + 2])

    centre = np.array([40.0, -25.0, 10.0])
    for _ in range(300):  # millimetre triangles 40 units from the origin
        p = centre + rng.uniform(-0.5, 0.5, 3)
        add(p, p + rng.normal(size=3) * 1e-3, p + rng.normal(size=3) * 1e-3)
    for _ in range(100):  # slivers: 1 long, 1e-4 wide
        p = centre + rng.uniform(-0.5, 0.5, 3)
        e = rng.normal(size=3)
        e /= np.linalg.norm(e)
        f = np.cross(e, rng.normal(size=3))
        f /= np.linalg.norm(f)
        add(p, p + e, p + e * 0.5 + f * 1e-4)
    for _ in range(20):  # degenerate: collinear and coincident vertices (never hit)
        p = centre + rng.uniform(-0.5, 0.5, 3)
        e = rng.normal(size=3)
        add(p, p + e, p + 2 * e)
        add(p, p, p + e)
    for _ in range(60):  # ordinary triangles as a backdrop
        p = centre + rng.uniform(-0.6, 0.6, 3)
        add(p, p + rng.normal(size=3) * 0.3, p + rng.normal(size=3) * 0.3)
    return np.array(verts, np.float32), np.array(idx, np.uint32), centre, rng


def quad_scene():
    """Parallelograms as two triangles each, in every combination of diagonal, winding and corner order, with
    shared or duplicated vertices, near the origin and 40 units away; plus pairs that only look like quads
    (trapezoids, kinked quads, a corner moved off the parallelogram) and single triangles in between.
    Returns (V, I, centre, rng, n_quads): the first 2*n_quads triangles are the true parallelograms."""
    rng = np.random.default_rng(23)
    verts, idx = [], []

    def tri_pair(corners, diagonal, rot_a, rot_b, flip_a, flip_b, share):
        p0, p1, p2, p3 = corners
        a, b = ([0, 1, 2], [0, 2, 3]) if diagonal == 0 else ([0, 1, 3], [1, 2, 3])
        a = a[rot_a:] + a[:rot_a]
        b = b[rot_b:] + b[:rot_b]
        if flip_a:
            a = a[::-1]
        if flip_b:
            b = b[::-1]
        base = len(verts)
        if share:
            verts.extend([p0, p1, p2, p3])
            idx.append([base + k for k in a])
            idx.append([base + k for k in b])
        else:
            verts.extend([corners[k] for k in a] + [corners[k] for k in b])
            idx.append([base, base + 1, base + 2])
            idx.append([base + 3, base + 4, base + 5])

    centre = np.array([40.0, -25.0, 10.0])
    n_quads = 0
    for where in (np.zeros(3), centre):
        for diagonal in (0, 1):
            for rot_a in range(3):
                for rot_b in range(3):
                    for flips in range(4):
                        p0 = (where + rng.uniform(-0.5, 0.5, 3)).astype(np.float32)
                        e1 = (rng.normal(size=3) * 0.3).astype(np.float32)
                        e2 = (rng.normal(size=3) * 0.3).astype(np.float32)
                        p1 = p0 + e1
                        p3 = p0 + e2
                        p2 = p1 + e2  # float32 sums: the fourth corner is off by the rounding of the coordinates
                        tri_pair([p0, p1, p2, p3], diagonal, rot_a, rot_b, flips & 1, flips >> 1, (rot_a + rot_b) & 1)
                        n_quads += 1
    for k in range(60):  # look-alikes that must stay two triangles
        p0 = centre + rng.uniform(-0.5, 0.5, 3)
        e1, e2 = rng.normal(size=3) * 0.3, rng.normal(size=3) * 0.3
        kind = k % 3
        if kind == 0:    # trapezoid: planar, not a parallelogram
            corners = [p0, p0 + e1, p0 + 0.6 * e1 + e2, p0 + e2]
        elif kind == 1:  # kinked: fourth corner out of the plane
            corners = [p0, p0 + e1, p0 + e1 + e2 + 0.05 * np.cross(e1, e2) / np.linalg.norm(np.cross(e1, e2)), p0 + e2]
        else:            # parallelogram with one corner moved by 1e-4
            corners = [p0, p0 + e1, p0 + e1 + e2 + 1e-4, p0 + e2]
        tri_pair([np.asarray(c, np.float32) for c in corners], k & 1, 0, 0, 0, 0, 1)
    for _ in range(40):  # singles
        p = centre + rng.uniform(-0.6, 0.6, 3)
        base = len(verts)
        verts.extend([p, p + rng.normal(size=3) * 0.3, p + rng.normal(size=3) * 0.3])
        idx.append([base, base + 1, base + 2])
    return np.array(verts, np.float32), np.array(idx, np.uint32), centre, rng, n_quads


def quad_rays(V, I, rng, n=8000):
    """Rays from around either cluster aimed at points of random triangles: interiors, the shared diagonals,
    outer edges and corners."""
    tri = rng.integers(0, len(I), n)
    w = rng.dirichlet([0.4, 0.4, 0.4], n)
    k = n // 4
    w[:k] = 0.0  # exactly on an edge: two corners share the weight
    e = rng.integers(0, 3, k)
    u = rng.uniform(0, 1, k)
    w[np.arange(k), e] = u
    w[np.arange(k), (e + 1) % 3] = 1 - u
    target = V[I[tri, 0]] * w[:, :1] + V[I[tri, 1]] * w[:, 1:2] + V[I[tri, 2]] * w[:, 2:3]
    o = (target + rng.normal(size=(n, 3)) * 1.5).astype(np.float32)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def awkward_rays(V, I, centre, rng):
    n = 6000
    o = (centre + rng.normal(size=(n, 3)) * 1.5).astype(np.float32)
    # aim at points of random triangles (their interiors, edges and vertices), so most rays hit something
    tri = rng.integers(0, len(I), n)
    w = rng.dirichlet([0.3, 0.3, 0.3], n)  # mass near edges and corners
    w[: n // 6] = np.eye(3)[rng.integers(0, 3, n // 6)]  # exactly at vertices
    target = (V[I[tri, 0]] * w[:, :1] + V[I[tri, 1]] * w[:, 1:2] + V[I[tri, 2]] * w[:, 2:3])
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    return o, d


def check_against_float64(V, I, o, d, prim, t, b1, b2):
    """Asserts that (prim, t, b1, b2) is a sound closest-hit answer for every ray."""
    n = len(o)
    # ground truth in float64: per ray and triangle, the plane hit and its barycentrics.  A triangle is
    # a definite hit (miss) when every barycentric is above (one is below) a margin that covers the
    # float32 rounding of the inputs relative to the triangle's height; in between either answer is right.
    V64, o64, d64 = V.astype(np.float64), o.astype(np.float64), d.astype(np.float64)
    A, B, Cc = V64[I[:, 0]], V64[I[:, 1]], V64[I[:, 2]]
    e1, e2 = B - A, Cc - A
    nrm = np.cross(e1, e2)
    area2 = np.linalg.norm(nrm, axis=1)
    longest = np.maximum(np.maximum(np.linalg.norm(e1, axis=1), np.linalg.norm(e2, axis=1)), np.linalg.norm(Cc - B, axis=1))
    height = np.where(longest > 0, area2 / np.maximum(longest, 1e-300), 0.0)
    margin = np.where(height > 0, 16 * 2.0 ** -23 * 60.0 / np.maximum(height, 1e-300), np.inf)  # coordinates < 60
    denom = d64 @ nrm.T  # rays x triangles
    with np.errstate(divide="ignore", invalid="ignore"):
        tt = ((A[None] - o64[:, None]) * nrm[None]).sum(-1) / denom
        Pt = o64[:, None] + tt[..., None] * d64[:, None]
        rel = Pt - A[None]
        nn = (nrm * nrm).sum(-1)
        bb1 = (np.cross(rel, e2[None]) * nrm[None]).sum(-1) / nn
        bb2 = (np.cross(e1[None], rel) * nrm[None]).sum(-1) / nn
    lo = np.minimum(np.minimum(bb1, bb2), 1 - bb1 - bb2)
    front = tt > 1e-6
    definite_hit = front & (lo > margin[None]) & np.isfinite(tt)
    definite_miss = ~(front & (lo > -margin[None]) & np.isfinite(tt))
    strict_miss = definite_miss & np.isfinite(margin)[None]  # the oracle's loop may accept an exactly collinear triangle
    t_first = np.where(definite_hit, tt, np.inf).min(axis=1)
    hit = prim >= 0
    rows = np.arange(n)
    assert not definite_miss[rows[hit], prim[hit]].any()          # never reports a triangle it cannot have hit
    assert (t[hit] <= t_first[hit] * (1 + 1e-4) + 1e-4).all()       # never skips a certain, nearer hit
    assert not np.isfinite(t_first[~hit]).any()                    # never misses a certain hit
    cosang = np.abs(denom[rows[hit], prim[hit]]) / np.maximum(area2[prim[hit]], 1e-300)
    t_tol = 1e-4 + 16 * 2.0 ** -23 * 60.0 / np.maximum(cosang, 1e-12)  # grazing planes are ill-conditioned in t
    dt = np.abs(t[hit] - tt[rows[hit], prim[hit]])
    well_defined = margin[prim[hit]] < 0.5  # not the (nearly) collinear ones, whose plane is rounding noise
    assert (dt < t_tol)[well_defined].all()  # distance of the reported triangle
    assert hit.mean() > 0.5
    # barycentrics of the reported triangle (corner order of that triangle, not of a neighbour)
    db = np.maximum(np.abs(b1[hit] - bb1[rows[hit], prim[hit]]), np.abs(b2[hit] - bb2[rows[hit], prim[hit]]))
    assert (db < margin[prim[hit]] + 1e-4)[well_defined].all()
    assert (b1[hit] >= -1e-3).all() and (b2[hit] >= -1e-3).all() and (b1[hit] + b2[hit] <= 1 + 1e-3).all()
    return definite_miss, strict_miss, t_first, rows
