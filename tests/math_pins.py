"""Expected values of the reference's estimator by deterministic float64 quadrature.

The reference holds no numeric vectors for its integrator, so the CPU oracle and the GPU kernels are pinned
here against mathematics instead: every function below integrates the formulas of optix_device.cu on a grid
(no random numbers, no code shared with oracle/ or the product) and the tests compare Monte-Carlo means of
the oracle and of the HIP kernels with these numbers.

Scene constants (optix_wrapper.cpp:386-464, 842-845): the floor is white (0.789, 0.801, 0.800), Roughness
0.25, IOR 1.5; the rectangle light used by next-event estimation has centre (0, 0, 1.99), half extents 0.3
along x and y, normal (0, 0, -1) and emits 15.
"""
import numpy as np

PI = np.pi
LIGHT_POS = np.array([0.0, 0.0, 1.99])
LIGHT_HALF = (0.3, 0.3)
LIGHT_EMIT = 15.0
LIGHT_N = np.array([0.0, 0.0, -1.0])
WHITE = np.array([0.789, 0.801, 0.800])
ROUGHNESS, IOR = 0.25, 1.5
ALPHA = max(ROUGHNESS * ROUGHNESS, 1e-4)  # optix_device.cu:488
F0 = ((IOR - 1.0) / (IOR + 1.0)) ** 2     # optix_device.cu:484


def schlick(c):
    return F0 + (1.0 - F0) * np.maximum(0.0, 1.0 - c) ** 5


def ggx_d(noh):
    a2 = ALPHA * ALPHA
    den = noh * noh * (a2 - 1.0) + 1.0
    return a2 / (PI * den * den)


def g1(c):
    a2 = ALPHA * ALPHA
    return 2.0 * c / (c + np.sqrt(a2 + (1.0 - a2) * c * c))


def opaque_constants(n, v):
    """spec_prob, diff_prob, kd of optix_device.cu:484-492 for normal n and view vector v (towards the eye)."""
    nov = max(float(n @ v), 0.0)
    f = schlick(nov)
    spec_prob = min(0.99, max(0.01, f))
    return nov, spec_prob, 1.0 - spec_prob, WHITE * (1.0 - f)


def expected_nee(p, n, v, nodes=96):
    """E over the two light-sample draws of the MIS-weighted next-event term Lo of optix_device.cu:495-548 at an
    unshadowed point p with geometric normal n, seen from direction v: Gauss-Legendre over the light rectangle."""
    nov, spec_prob, diff_prob, kd = opaque_constants(n, v)
    x, wq = np.polynomial.legendre.leggauss(nodes)
    uu, vv = np.meshgrid(x, x, indexing="ij")
    ww = np.outer(wq, wq) / 4.0  # u, v uniform on [-1, 1]^2: density 1/4
    lp = LIGHT_POS[None, None, :] + np.stack([uu * LIGHT_HALF[0], vv * LIGHT_HALF[1], 0 * uu], -1)
    L = lp - p
    dist2 = np.maximum((L * L).sum(-1), 1e-6)
    dist = np.sqrt(dist2)
    wi = L / dist[..., None]
    cos_s = np.maximum(0.0, wi @ n)
    cos_l = np.maximum(0.0, -(wi @ LIGHT_N))
    h = v[None, None, :] + wi
    h /= np.linalg.norm(h, axis=-1, keepdims=True)
    voh = np.maximum(h @ v, 0.0)
    noh = np.maximum(h @ n, 0.0)
    d = ggx_d(noh)
    g1v, g1l = g1(nov), g1(cos_s)
    f_spec = schlick(voh) * d * g1v * g1l / np.maximum(4.0 * nov * cos_s, 1e-6)
    pdf_bsdf = d * noh * g1v / np.maximum(4.0 * voh, 1e-6) * spec_prob + cos_s / PI * diff_prob
    area = 4.0 * LIGHT_HALF[0] * LIGHT_HALF[1]
    pdf_light = dist2 / (cos_l * area)
    w = pdf_light / (pdf_light + pdf_bsdf)
    geom = cos_s * cos_l / dist2 * area
    lo = LIGHT_EMIT * (kd[None, None, :] / PI + f_spec[..., None]) * (geom * w)[..., None]
    lo = np.where(((cos_s > 0) & (cos_l > 0))[..., None], lo, 0.0)
    return (lo * ww[..., None]).sum((0, 1))


def expected_throughput_factor(n, v, n_theta=6000, n_phi=1440):
    """E over the lobe choice and the lobe's two draws of the factor the throughput is multiplied with at an
    opaque hit (optix_device.cu:564-658, unguided).

    Diffuse branch (probability diff_prob): kd * cos / (pi * (cos / pi) * diff_prob) = kd / diff_prob, so it
    contributes kd.  Specular branch (probability spec_prob): F D G1V G1L / (4 NoV NoL) * NoL / (pdf * spec_prob)
    with the pdf the reference computes, D NoH G1V / (4 VoH), while the half vector is drawn from the
    distribution of visible normals, G1V max(VoH, 0) D / NoV per solid angle of H; the integral over H of
    weight x density is evaluated on a (theta_h, phi_h) midpoint grid."""
    nov, spec_prob, diff_prob, kd = opaque_constants(n, v)
    t = np.array([1.0, 0.0, 0.0]) if abs(n[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
    t = t - n * (t @ n)
    t /= np.linalg.norm(t)
    b = np.cross(n, t)
    # D is concentrated within a few alpha of the normal: theta_h = pi/2 * s^3 puts the grid where the mass is
    s = (np.arange(n_theta) + 0.5) / n_theta
    th = 0.5 * PI * s ** 3
    dth = 0.5 * PI * 3.0 * s * s / n_theta
    ph = (np.arange(n_phi) + 0.5) / n_phi * 2.0 * PI
    st, ct = np.sin(th)[:, None], np.cos(th)[:, None]
    h = st[..., None] * (np.cos(ph)[None, :, None] * t + np.sin(ph)[None, :, None] * b) + ct[..., None] * n
    voh = h @ v
    l = 2.0 * voh[..., None] * h - v
    nol = l @ n
    noh = ct + 0 * voh
    ok = (voh > 0) & (nol > 0)
    vohc, nolc = np.maximum(voh, 1e-12), np.maximum(nol, 1e-12)
    d = ggx_d(noh)
    g1v, g1l = g1(nov), g1(nolc)
    brdf = schlick(vohc) * d * g1v * g1l / np.maximum(4.0 * nov * nolc, 1e-6)
    pdf_code = d * noh * g1v / np.maximum(4.0 * vohc, 1e-6)
    weight = brdf * nolc / np.maximum(pdf_code * spec_prob, 1e-6)
    density_h = g1v * vohc * d / nov
    integrand = np.where(ok, weight * density_h, 0.0) * st * (dth[:, None] * (2.0 * PI / n_phi))
    e_spec = integrand.sum()
    norm = (np.where(voh > 0, density_h, 0.0) * st * (dth[:, None] * (2.0 * PI / n_phi))).sum()
    return spec_prob * e_spec + kd, norm  # norm: the visible-normal density integrates to 1 (quadrature check)


def vndf_bin_probabilities(v_local, theta_edges, n_phi_bins, sub=40):
    """Probability that a half vector drawn from the distribution of visible normals (shading frame, normal =
    +z, view vector v_local) falls into each (theta_h, phi_h) bin: the density integrated on a sub-grid."""
    nov = v_local[2]
    probs = np.zeros((len(theta_edges) - 1, n_phi_bins))
    for i in range(len(theta_edges) - 1):
        th = theta_edges[i] + (np.arange(sub * 4) + 0.5) / (sub * 4) * (theta_edges[i + 1] - theta_edges[i])
        dth = (theta_edges[i + 1] - theta_edges[i]) / (sub * 4)
        for j in range(n_phi_bins):
            ph = (j + (np.arange(sub) + 0.5) / sub) * 2.0 * PI / n_phi_bins
            dph = 2.0 * PI / n_phi_bins / sub
            st, ct = np.sin(th)[:, None], np.cos(th)[:, None]
            hx, hy, hz = st * np.cos(ph)[None, :], st * np.sin(ph)[None, :], ct + 0 * ph[None, :]
            voh = hx * v_local[0] + hy * v_local[1] + hz * v_local[2]
            dens = g1(nov) * np.maximum(voh, 0.0) * ggx_d(hz) / nov
            probs[i, j] = (dens * st).sum() * dth * dph
    return probs


def vmf_cdf(t, kappa):
    """P(mu . omega <= t) of the von Mises-Fisher distribution kappa / (4 pi sinh kappa) exp(kappa mu . omega)."""
    return (np.exp(kappa * (t - 1.0)) - np.exp(-2.0 * kappa)) / (1.0 - np.exp(-2.0 * kappa))


def vmf_pdf(mu, kappa, w):
    """kappa / (4 pi sinh kappa) exp(kappa mu . w) in the form that stays finite (guiding_gpu.cuh:80-84)."""
    return kappa / (2.0 * PI * (1.0 - np.exp(-2.0 * kappa))) * np.exp(kappa * (w @ mu - 1.0))


def expected_guided_throughput_factor(n, v, lobes, n_theta=4000, n_phi=1024):
    """E of the throughput factor at an opaque hit whose region holds the given lobes [(mu, kappa, weight, rgb), ...],
    with the field on (optix_device.cu:596-658).  The diffuse branch (probability diff_prob) picks lobe i with
    probability weight_i / total (guiding_gpu.cuh:86-106), draws a direction w from it, then keeps w with
    probability a = p_g / (p_g + p_b), where p_g is the pdf of the whole mixture at w and p_b = max(cos, 0) / pi --
    the reference's rule, which is not an unbiased one-sample MIS -- multiplying the throughput by
    kd * rgb_i * cos / (pi * p_g * diff_prob); otherwise it takes a cosine-distributed direction with factor
    kd / diff_prob.  So the branch contributes
        kd * Sum_i weight_i / total * Integral vmf_i(w) [ a rgb_i cos / (pi p_g) + (1 - a) ] dw,
    each integral on a (theta, phi) grid about mu_i whose cells carry equal shares of lobe i's mass; the specular
    branch is expected_throughput_factor's."""
    nov, spec_prob, diff_prob, kd = opaque_constants(n, v)
    e_unguided, _ = expected_throughput_factor(n, v, n_theta=3000, n_phi=720)
    e_spec_part = e_unguided - kd  # spec_prob * e_spec
    total = sum(l[2] for l in lobes)
    mus = [np.asarray(l[0], np.float64) / np.linalg.norm(l[0]) for l in lobes]

    def mixture(w):
        return sum(l[2] / total * vmf_pdf(m, l[1], w) for l, m in zip(lobes, mus))

    e_diff = np.zeros(3)
    for (_, kappa, weight, rgb), mu in zip(lobes, mus):
        t = np.array([1.0, 0.0, 0.0]) if abs(mu[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
        t = t - mu * (t @ mu)
        t /= np.linalg.norm(t)
        b = np.cross(mu, t)
        u = (np.arange(n_theta) + 0.5) / n_theta
        ct = 1.0 + np.log(np.exp(-2.0 * kappa) + (1.0 - np.exp(-2.0 * kappa)) * u) / kappa  # inverse cdf: equal-mass rings
        st = np.sqrt(np.maximum(0.0, 1.0 - ct * ct))
        ph = (np.arange(n_phi) + 0.5) / n_phi * 2.0 * PI
        w = st[:, None, None] * (np.cos(ph)[None, :, None] * t + np.sin(ph)[None, :, None] * b) + ct[:, None, None] * mu
        cos = np.maximum(w @ n, 0.0)
        p_g = mixture(w)
        p_b = cos / PI
        a = p_g / (p_g + p_b)
        inner = a[..., None] * np.asarray(rgb, np.float64)[None, None, :] * (cos / (PI * p_g))[..., None] + (1.0 - a)[..., None]
        e_diff += weight / total * inner.mean((0, 1))
    return e_spec_part + kd * e_diff


def chi2_threshold(dof):
    """A bound a chi-square statistic with `dof` degrees of freedom exceeds with probability about 1e-5
    (Wilson-Hilferty)."""
    z = 4.27
    return dof * (1.0 - 2.0 / (9.0 * dof) + z * np.sqrt(2.0 / (9.0 * dof))) ** 3


def camera_ray(px, py, width, height):
    """Centre ray of pixel (px, py) of the built-in camera: eye (0, 3, 1) looking at (0, 0, 1), up +z, vertical
    field of view 45 degrees (optix_wrapper.cpp:813-833, optix_device.cu:890-900)."""
    eye = np.array([0.0, 3.0, 1.0])
    w = np.array([0.0, -1.0, 0.0])
    u_axis = np.cross(w, [0.0, 0.0, 1.0])
    u_axis /= np.linalg.norm(u_axis)
    v_axis = np.cross(u_axis, w)
    th = np.tan(np.radians(45.0) / 2)
    fx, fy = (px + 0.5) / width, (py + 0.5) / height
    d = w + (2 * fx - 1) * th * (width / height) * u_axis + (1 - 2 * fy) * th * v_axis
    return eye, d / np.linalg.norm(d)


def expected_depth_one_pixel(px, py, width, height):
    """Expected value of a pixel whose centre ray lands on an unshadowed part of the floor, rendered with
    max_depth = 1: the one bounce contributes (next-event term) x (throughput after the bounce), no roulette
    (optix_device.cu:660-672 ends the path first), and the camera ray carries throughput 1."""
    eye, d = camera_ray(px, py, width, height)
    t = -eye[2] / d[2]
    p = eye + t * d
    assert d[2] < 0 and abs(p[0]) < 1 and abs(p[1]) < 1
    n = np.array([0.0, 0.0, 1.0])
    e_w, norm = expected_throughput_factor(n, -d, n_theta=3000, n_phi=720)
    assert abs(norm - 1.0) < 4e-3
    return expected_nee(p, n, -d) * e_w, p
